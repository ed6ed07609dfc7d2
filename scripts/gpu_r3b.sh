#!/bin/bash
# Round 3, GPU call B: parity of the pruned library (+ one-wave k_reproject_fix), timing A/B against the skip-loop probe, VALU counts
# of both (how much of k_integrate is the per-item fixed part: queue, slot, loads, culling preamble, stores?), kernel stats.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03b; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
timeout 900 python -m pytest tests/test_tsdf_gpu.py tests/test_host_programs_gpu.py tests/test_icp_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/pytest_gpu_$TAG.log; tail -6 gpurun_out/pytest_gpu_$TAG.log
echo "== t=${SECONDS}s A/B"
bash scripts/ab_libs.sh 2 main skiploop > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== t=${SECONDS}s stats"
bash scripts/gpu_prof.sh $TAG --steps 20 --warmup 1 --cpu-sample 0 --icp-pairs 0 --no-streamed --no-alone --min-seconds 0.01 > /dev/null 2>&1; python scripts/kstats.py gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv 2>&1 | head -8
echo "== t=${SECONDS}s pmc"
for v in main skiploop; do
  if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$R/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
  OUT=$R/gpurun_out/pmc_${TAG}_$v; mkdir -p $OUT; cd /tmp; i=0
  for CS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d /tmp/pmc_${TAG}_${v}_$i -o p$i -- python $R/bench.py --steps 20 --warmup 1 --cpu-sample 0 --icp-pairs 0 --no-streamed --no-alone --min-seconds 0.01 > $OUT/run_$i.log 2>&1
    for f in $(find /tmp/pmc_${TAG}_${v}_$i -name "*counter_collection.csv"); do cp "$f" $OUT/pass${i}_counter_collection.csv; done
  done
  cd $R; head -2 $OUT/pass1_counter_collection.csv | cut -c1-600 > $OUT/csv_head.txt; python scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; rm -f $OUT/pass*_counter_collection.csv
  echo "=== $v"; grep -A9 "^k_integrate" $OUT/summary.txt; grep -A9 "^k_reproject_scatter" $OUT/summary.txt | head -10; grep -A9 "^k_prepare" $OUT/summary.txt | head -10
done
unset ER_HIP_LIB
echo "== done t=${SECONDS}s"
