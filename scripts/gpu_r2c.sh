#!/bin/bash
# Round 2, GPU call C: TSDF GPU tests with the stage-wise tiered kernel + the new multi-GPU program tests, A/B of the
# Reproject variants, kernel stats per variant, one PMC pass (VALU instructions) for the main build.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="${1:-r02c}"; mkdir -p gpurun_out
SECONDS=0
timeout 600 python -m pytest tests/test_tsdf_gpu.py tests/test_host_programs_gpu.py -m gpu -q --tb=short -p no:cacheprovider -n 4 > gpurun_out/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/pytest_gpu_$TAG.log; tail -25 gpurun_out/pytest_gpu_$TAG.log
echo "== t=${SECONDS}s A/B"
bash scripts/ab_libs.sh 2 main rt8 rt4n2 exactreproj > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== t=${SECONDS}s stats"
for v in main rt8 rt4n2; do
  if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$R/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
  bash scripts/gpu_prof.sh ${TAG}_$v --steps 20 --warmup 2 --cpu-sample 0 --icp-pairs 0 --no-streamed --min-seconds 0.2 > /dev/null 2>&1; echo "-- $v"; python scripts/kstats.py gpurun_out/prof_${TAG}_$v/${TAG}_${v}_kernel_stats.csv 2>&1 | head -4
done
unset ER_HIP_LIB
echo "== t=${SECONDS}s pmc"
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --output-format csv -d /tmp/pmc_${TAG}_1 -o p1 -- python $R/bench.py --steps 4 --warmup 1 --cpu-sample 0 --icp-pairs 0 --no-streamed --min-seconds 0.01 > $OUT/run_1.log 2>&1
for f in $(find /tmp/pmc_${TAG}_1 -name "*counter_collection.csv"); do cp "$f" $OUT/pass1_counter_collection.csv; done
cd $R; python scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; grep -A9 "^k_reproject_tiered\|^k_integrate" $OUT/summary.txt | head -40
echo "== done t=${SECONDS}s"
