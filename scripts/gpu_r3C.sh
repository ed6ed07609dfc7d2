#!/bin/bash
# Round 3, GPU call C2: when do the persistent workgroups of k_integrate finish, and how long is the longest item?  (patched variant 'dbg')
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
ER_HIP_LIB=$R/elasticreconstruction_amd/_ab/liber_hip_dbg.so timeout 300 python scripts/item_time_probe.py 2>&1 | tail -20 | tee gpurun_out/item_time_r03C.txt
