#!/bin/bash
# Round 3, GPU call Z: which part of the new item loop costs 40 %?  main = plan records + claim in flight + row prefetch; norow = without the row
# prefetch; noclaim = without either (plan records only, claim at the end of the item); base = the plain queue loop.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03z1; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
AB_ALONE=1 bash scripts/ab_libs.sh 2 main norow noclaim base > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
