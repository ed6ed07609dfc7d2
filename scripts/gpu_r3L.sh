#!/bin/bash
# Round 3, GPU call L: the library's own Cholesky (recursive blocked potrf_lower on a 64 x 64 LDS kernel + rocblas_dtrsm / dsyrk; rocSOLVER removed):
# FragmentOptimizer tests, the dense system at 20 fragments (43 740 unknowns), and the multi-process stress that used to trip rocsolver_dpotrf.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03L; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
timeout 900 python -m pytest tests/test_fopt_gpu.py tests/test_host_programs_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -5
echo "== t=${SECONDS}s scale probe (20 fragments)"
timeout 600 python scripts/fopt_scale_probe.py 20 2>&1 | tail -4
echo "== t=${SECONDS}s stress: 160 runs, 4 at a time, 2 background benches"
timeout 900 python scripts/gpu_fopt_flake.py 160 2 4 2>&1 | tail -8 | tee gpurun_out/fopt_flake_$TAG.txt
echo "== done t=${SECONDS}s"
