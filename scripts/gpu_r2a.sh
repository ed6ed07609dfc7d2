#!/bin/bash
# Round 2, first GPU call: -m gpu suite (main library), the TSDF parity tests again with the -DER_FAST_CULL variant,
# the default bench (driver flags), kernel-trace stats, one interleaved A/B main vs fastcull.
# usage: bash scripts/gpu_r2a.sh <tag>
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="${1:-r02a}"; mkdir -p gpurun_out
SECONDS=0
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 4 > gpurun_out/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/pytest_gpu_$TAG.log; tail -15 gpurun_out/pytest_gpu_$TAG.log
echo "== t=${SECONDS}s fastcull parity"
ER_HIP_LIB=$R/elasticreconstruction_amd/_ab/liber_hip_fastcull.so timeout 400 python -m pytest tests/test_tsdf_gpu.py tests/test_host_programs_gpu.py -m gpu -q --tb=short -p no:cacheprovider -n 4 > gpurun_out/pytest_fastcull_$TAG.log 2>&1
echo "pytest(fastcull) exit $? after ${SECONDS}s" >> gpurun_out/pytest_fastcull_$TAG.log; tail -6 gpurun_out/pytest_fastcull_$TAG.log
echo "== t=${SECONDS}s smoke"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke_$TAG.log; tail -3 gpurun_out/smoke_$TAG.log
echo "== t=${SECONDS}s bench"
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err; echo "bench exit $?"; tail -1 gpurun_out/bench_default_$TAG.json | cut -c1-1500; tail -5 gpurun_out/bench_default_$TAG.err
echo "== t=${SECONDS}s A/B"
bash scripts/ab_libs.sh 2 main fastcull > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== t=${SECONDS}s stats"
bash scripts/gpu_prof.sh $TAG --steps 20 --warmup 2 --cpu-sample 0 --icp-pairs 0 --no-streamed --min-seconds 0.2 > /dev/null 2>&1; python scripts/kstats.py gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv 2>&1 | head -14
echo "== done t=${SECONDS}s"
