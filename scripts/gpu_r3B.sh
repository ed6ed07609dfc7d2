#!/bin/bash
# Round 3, GPU call B2: every wave of k_integrate claims its own 8 x 8 x 8 cube (main: no workgroup barrier) against the 4-wave items (planrec).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03B; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
timeout 1200 python -m pytest tests/test_tsdf_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -5
echo "== t=${SECONDS}s A/B"
AB_ALONE=1 bash scripts/ab_libs.sh 3 main planrec > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
