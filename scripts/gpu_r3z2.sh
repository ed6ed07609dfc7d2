#!/bin/bash
# Round 3, GPU call Z2: one work queue per XCD (main) against the single queue with per-unit plan records (planrec) and against the round's
# previous kernel (base: four dependent look-ups per item, slot allocation in k_integrate).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03z2; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
timeout 1200 python -m pytest tests/test_tsdf_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -15
echo "== t=${SECONDS}s A/B"
AB_ALONE=1 bash scripts/ab_libs.sh 3 main planrec base > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
