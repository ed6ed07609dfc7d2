#!/bin/bash
# round 4, call 7: explicitly global pointers in the hot loads of path B (main) against the flat loads the descriptors give by default
# (variant flatptr); fused entry with 2 / 3 / 4 / 6 shares; er_cloud_create_batch on two lanes; the path-B tests on main.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_icp_gpu.py -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/pytest_icp_r04g.log 2>&1; echo "pytest icp exit $?"; tail -4 gpurun_out/pytest_icp_r04g.log
for rep in 1 2; do
  for v in main flatptr; do
    if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$PWD/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
    echo "== $v"; ER_PROBE_SHARES=$([ $rep = 1 ] && echo "2 3 4 6" || echo "3") ER_PROBE_CLOUDS=$([ $rep = 1 ] && echo 1 || echo 0) timeout 300 python scripts/icp_list_probe.py 50 12 2>&1 | tail -8
  done
done
