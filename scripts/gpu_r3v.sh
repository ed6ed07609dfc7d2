#!/bin/bash
# Round 3, GPU call V: where do the waves of k_integrate spend their time?  SQ wait / active counters (separate passes, kernel-trace only).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03v; export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT; cd /tmp
SECONDS=0
i=0
for CS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
          "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU" \
          "SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE" \
          "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d /tmp/pmc_${TAG}_$i -o p$i -- python $R/bench.py --steps 20 --warmup 1 --cpu-sample 0 --icp-pairs 0 --no-streamed --no-alone --other-configs 0 --min-seconds 0.01 > $OUT/run_$i.log 2>&1
  for f in $(find /tmp/pmc_${TAG}_$i -name "*counter_collection.csv"); do cp "$f" $OUT/pass${i}_counter_collection.csv; done
  echo "pass $i done t=${SECONDS}s"
done
cd $R; python scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; grep -A40 "^k_integrate" $OUT/summary.txt | head -45
rm -f $OUT/pass*_counter_collection.csv
echo "== done t=${SECONDS}s"
