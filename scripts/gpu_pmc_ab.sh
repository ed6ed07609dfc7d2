#!/bin/bash
# PMC comparison of library variants (separate rocprofv3 --pmc passes, kernel-trace only): scripts/gpu_pmc_ab.sh TAG variant...
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="$1"; shift; mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --list-avail > gpurun_out/counters_avail_$TAG.txt 2>&1
SECONDS=0
for v in "$@"; do
  if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$R/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
  OUT=$R/gpurun_out/pmc_${TAG}_$v; mkdir -p $OUT; cd /tmp
  i=0
  for CS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
            "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
            "SQ_IFETCH SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
            "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d /tmp/pmc_${TAG}_${v}_$i -o p$i -- python $R/bench.py --steps 20 --warmup 1 --cpu-sample 0 --icp-pairs 0 --no-streamed --no-alone --min-seconds 0.01 > $OUT/run_$i.log 2>&1
    for f in $(find /tmp/pmc_${TAG}_${v}_$i -name "*counter_collection.csv"); do cp "$f" $OUT/pass${i}_counter_collection.csv; done
    echo "$v pass $i done t=${SECONDS}s"
  done
  cd $R; python scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; rm -f $OUT/pass*_counter_collection.csv
  echo "=== $v"; grep -A40 "^k_integrate" $OUT/summary.txt | awk '/^k_[a-z_]*/{n++} n<2{print}'
done
