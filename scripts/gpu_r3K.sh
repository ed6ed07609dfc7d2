#!/bin/bash
# Round 3, GPU call K: order of the units in the work queue of k_integrate: by cost with ties in arrival order (main), by cost with ties in key order
# (detplan: deterministic, spatial neighbours adjacent), by key only (keyorder: no cost order).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03K; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
AB_ALONE=1 bash scripts/ab_libs.sh 3 main detplan keyorder > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
