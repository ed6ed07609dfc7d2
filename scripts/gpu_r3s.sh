#!/bin/bash
# Round 3, GPU call S: float32 first try of the unit key in k_prepare (main) against the float64-only form (k64); probe: stage 3 of
# Reproject with fused multiply-adds (fmaprobe, unguarded = an upper bound of what a guarded version could gain).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03s; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
timeout 900 python -m pytest tests/test_tsdf_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -3
echo "== t=${SECONDS}s A/B"
bash scripts/ab_libs.sh 3 main k64 fmaprobe > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
