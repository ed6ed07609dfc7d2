#!/bin/bash
# Round 3, GPU call E2: on top of four rows per lane / four persistent workgroups per CU (main): five workgroups per CU (b5), a third pre-pass stream
# with a four-deep batch ring (d43), the float32 first try of the unit key in k_prepare (key32).  The pre-pass chain is what the voxel stream waits for now.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03E; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
AB_ALONE=1 bash scripts/ab_libs.sh 2 main b5 d43 key32 > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
