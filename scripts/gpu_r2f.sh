#!/bin/bash
# Round 2, GPU call F: the tests touched since call E, A/B of the k_integrate variants (branch-free finish vs branchy, cull first).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="${1:-r02f}"; mkdir -p gpurun_out
SECONDS=0
timeout 800 python -m pytest tests/test_tsdf_gpu.py tests/test_host_programs_gpu.py tests/test_fopt_gpu.py -m gpu -q --tb=short -p no:cacheprovider -n 4 > gpurun_out/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/pytest_gpu_$TAG.log; tail -30 gpurun_out/pytest_gpu_$TAG.log
echo "== t=${SECONDS}s A/B"
bash scripts/ab_libs.sh 3 main branchy cullfirst > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
