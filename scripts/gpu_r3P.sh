#!/bin/bash
# Round 3, GPU call P: stream priorities again, now that the voxel stream waits for the pre-pass chain (final four-row kernel):
# prhi = the two pre-pass streams at the highest priority, prlo = at the lowest, main = default.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03P; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
bash scripts/ab_libs.sh 3 main prhi prlo > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
