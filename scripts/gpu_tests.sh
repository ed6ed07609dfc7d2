#!/bin/bash
# usage: bash scripts/gpu_tests.sh [pytest args...]   (default: the whole -m gpu suite + smoke)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider "$@" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -60 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -5 gpurun_out/smoke.log
