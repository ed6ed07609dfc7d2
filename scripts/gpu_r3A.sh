#!/bin/bash
# Round 3, GPU call A2: the item barrier of k_integrate.  main = one bare s_barrier per item behind an LDS wait (claim through two alternating LDS
# words); twobar = two bare barriers; rowlate = main + the voxel rows loaded after the culling preamble; planrec = two __syncthreads() (drain vmcnt).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03A; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
timeout 1200 python -m pytest tests/test_tsdf_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -5
echo "== t=${SECONDS}s A/B"
AB_ALONE=1 bash scripts/ab_libs.sh 3 main twobar rowlate planrec > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
