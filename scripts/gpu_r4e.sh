#!/bin/bash
# round 4, call 5: points per thread of k_icp_iter chosen per chunk (main) against the fixed 8 of round 3 (variant fixedpts): easy and hard list;
# er_cloud_create_batch (pageable and page-locked input); the path-B tests on main.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_icp_gpu.py -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/pytest_icp_r04e.log 2>&1; echo "pytest icp exit $?"; tail -4 gpurun_out/pytest_icp_r04e.log
for rep in 1 2; do
  for v in main fixedpts; do
    if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$PWD/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
    echo "== $v"; ER_PROBE_SHARES="3" ER_PROBE_CLOUDS=$([ $rep = 1 ] && echo 1 || echo 0) timeout 300 python scripts/icp_list_probe.py 50 12 2>&1 | tail -5
  done
done
