#!/bin/bash
# Generic GPU call: TSDF + host-program parity tests, interleaved A/B of library variants, kernel-trace stats of the in-tree build.
# usage: bash scripts/gpu_ab.sh TAG REPS variant1 variant2 ...     ("main" = in-tree liber_hip.so, others from elasticreconstruction_amd/_ab/)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="$1"; REPS="$2"; shift; shift; mkdir -p gpurun_out
SECONDS=0
timeout 800 python -m pytest tests/test_tsdf_gpu.py tests/test_host_programs_gpu.py -m gpu -q --tb=short -p no:cacheprovider -n 4 > gpurun_out/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/pytest_gpu_$TAG.log; tail -12 gpurun_out/pytest_gpu_$TAG.log
echo "== t=${SECONDS}s A/B"
bash scripts/ab_libs.sh $REPS "$@" > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== t=${SECONDS}s stats"
bash scripts/gpu_prof.sh $TAG --steps 20 --warmup 1 --cpu-sample 0 --icp-pairs 0 --no-streamed --no-alone --min-seconds 0.01 > /dev/null 2>&1; python scripts/kstats.py gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv 2>&1 | head -8
echo "== done t=${SECONDS}s"
