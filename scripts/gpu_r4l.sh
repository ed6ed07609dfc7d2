#!/bin/bash
# round 4, call 13: scheduler strategies for er_icp.hip (the flags go to every file of the variant; only path B is measured here)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for rep in 1 2; do
  for v in main icpmem icpilp icpnsmem icpiter; do
    if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$PWD/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
    echo "== $v"; ER_PROBE_SHARES="6" ER_PROBE_CLOUDS=0 ER_PROBE_HARD=$([ $rep = 1 ] && echo 1 || echo 0) timeout 300 python scripts/icp_list_probe.py 50 12 2>&1 | tail -3
  done
done
