#!/bin/bash
# round 4: SQ counters of the NN kernels with and without the compacted row tasks (two counter sets, each build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; R=$PWD
for v in main nocompact; do
  if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$R/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
  OUT=$R/gpurun_out/pmc_icp_r04w_$v; mkdir -p $OUT; cd /tmp; i=0
  for CS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    ER_PROBE_FUSED=0 ER_PROBE_CLOUDS=0 ER_PROBE_HARD=0 timeout 200 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d /tmp/pmc_r04w_${v}_$i -o p$i -- python $R/scripts/icp_list_probe.py 50 2 > $OUT/run_$i.log 2>&1
    for f in $(find /tmp/pmc_r04w_${v}_$i -name "*counter_collection.csv"); do cp "$f" $OUT/pass${i}_counter_collection.csv; done
    tail -1 $OUT/run_$i.log
  done
  cd $R; python scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; python scripts/icp_pmc_derive.py $OUT/summary.txt > $OUT/derived.txt 2>&1; cat $OUT/derived.txt
  rm -f $OUT/pass*_counter_collection.csv
done
