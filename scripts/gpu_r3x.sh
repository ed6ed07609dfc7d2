#!/bin/bash
# Round 3, GPU call X: per-unit plan records (slot allocation moved to k_plan) + claim and row prefetch of the next work item (main) against the plain queue loop (base).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03x; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
timeout 1200 python -m pytest tests/test_tsdf_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -15
echo "== t=${SECONDS}s A/B"
AB_ALONE=1 bash scripts/ab_libs.sh 3 main base > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
