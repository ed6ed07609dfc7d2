#!/bin/bash
# round 4: neighbour-row tasks without the empty ranges (ER_NN_COMPACT=1 ships) against the round-3 task list, and the two-bin variant
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_icp_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/pytest_icp_r04v.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/pytest_icp_r04v.log
for rep in 1 2; do
  for v in main nocompact bins; do
    if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$PWD/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
    echo "== $v"; ER_PROBE_FUSED=0 ER_PROBE_CLOUDS=0 ER_PROBE_HARD=0 timeout 300 python scripts/icp_list_probe.py 50 12 2>&1 | tail -1
  done
done
unset ER_HIP_LIB
echo "== main, all probes"; timeout 300 python scripts/icp_list_probe.py 50 12 2>&1 | tail -8
