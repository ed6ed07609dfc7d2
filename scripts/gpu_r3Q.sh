#!/bin/bash
# Round 3, GPU call Q2: ONE pre-pass stream (batch ring of 2 or 3) against the two pre-pass streams of the final pipeline (main).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03Q2; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
bash scripts/ab_libs.sh 2 main d21 d31 > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
