#!/bin/bash
# PMC passes (separate runs, kernel-trace only, as the MI355X guide prescribes) over the NN kernels of path B: the bench's 50-pair list
# through the three batch entry points (scripts/icp_list_probe.py), one rocprofv3 run per counter set.
#   usage: bash scripts/gpu_icp_pmc.sh <tag> [pairs] [reps]          -> gpurun_out/pmc_icp_<tag>/summary.txt
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="${1:-icp}"; PAIRS="${2:-50}"; REPS="${3:-3}"
OUT=$R/gpurun_out/pmc_icp_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for CS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
          "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
          "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
          "FETCH_SIZE WRITE_SIZE TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d /tmp/pmc_icp_${TAG}_$i -o p$i -- python $R/scripts/icp_list_probe.py $PAIRS $REPS > $OUT/run_$i.log 2>&1
  for f in $(find /tmp/pmc_icp_${TAG}_$i -name "*counter_collection.csv"); do cp "$f" $OUT/pass${i}_counter_collection.csv; done
  tail -1 $OUT/run_$i.log
done
cd $R
python scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -A22 -E "^k_count_inliers|^k_icp_iter|^k_find_corr" $OUT/summary.txt | head -90
rm -f $OUT/pass*_counter_collection.csv      # (tens of MB; the summary is what is kept)
