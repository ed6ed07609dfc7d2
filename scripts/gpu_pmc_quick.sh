#!/bin/bash
# One PMC pass (instruction counts + VALU busy) of the in-tree build or a variant: scripts/gpu_pmc_quick.sh TAG [variant]
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="$1"; v="${2:-main}"
if [ "$v" != main ]; then export ER_HIP_LIB=$R/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
OUT=$R/gpurun_out/pmcq_${TAG}_$v; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for CS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d /tmp/pmcq_${TAG}_${v}_$i -o p$i -- python $R/bench.py --steps 20 --warmup 1 --cpu-sample 0 --icp-pairs 0 --no-streamed --no-alone --min-seconds 0.01 > $OUT/run_$i.log 2>&1
  for f in $(find /tmp/pmcq_${TAG}_${v}_$i -name "*counter_collection.csv"); do cp "$f" $OUT/pass${i}_counter_collection.csv; done
done
cd $R; python scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; rm -f $OUT/pass*_counter_collection.csv
grep -A8 "^k_integrate\|^k_reproject_scatter\|^k_prepare" $OUT/summary.txt
