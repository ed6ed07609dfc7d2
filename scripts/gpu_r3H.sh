#!/bin/bash
# Round 3, GPU call H: a register diet for k_integrate by launch bounds (80 / 64 VGPRs with scratch spills) -- does the room it leaves to the pre-pass kernels pay?
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03H; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
AB_ALONE=1 bash scripts/ab_libs.sh 2 main mb6 mb6b6 mb8 > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
