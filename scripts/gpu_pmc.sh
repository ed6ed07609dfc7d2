#!/bin/bash
# PMC passes (separate runs, kernel-trace only, as the MI355X guide prescribes) of a short bench.py run.
# usage: bash scripts/gpu_pmc.sh <tag> [bench args...]
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="$1"; shift
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
i=0
for CS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
          "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
          "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d /tmp/pmc_${TAG}_$i -o p$i -- python $R/bench.py "$@" > $OUT/run_$i.log 2>&1
  for f in $(find /tmp/pmc_${TAG}_$i -name "*counter_collection.csv"); do cp "$f" $OUT/pass${i}_counter_collection.csv; done
done
cd $R
python scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
