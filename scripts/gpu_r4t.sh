#!/bin/bash
# round 4: candidates per trip of scan_range (ER_ICP_UNROLL: 4 ships)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for rep in 1 2; do
  for v in main un2 un3 un8; do
    if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$PWD/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
    echo "== $v"; ER_PROBE_FUSED=0 ER_PROBE_CLOUDS=0 ER_PROBE_HARD=0 timeout 300 python scripts/icp_list_probe.py 50 12 2>&1 | tail -1
  done
done
