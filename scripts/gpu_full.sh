#!/bin/bash
# Full validation: -m gpu suite, smoke, default bench (60 steps, CPU baseline), ICP section, rocprof stats.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="${1:-r01}"; mkdir -p gpurun_out
bash scripts/gpu_tests.sh
timeout 900 python bench.py > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err; tail -1 gpurun_out/bench_default_$TAG.json
bash scripts/gpu_prof.sh $TAG --steps 20 --warmup 2 --cpu-sample 0 --icp-pairs 0 > /dev/null; python scripts/kstats.py gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv
