#!/bin/bash
# round 4, call 9: scan_range as a tournament on 32-bit distance bits (main; noslp = the same without the SLP vectoriser's packed-math
# pairs) against the exact 64-bit scan of every candidate (exact, exactnoslp); the path-B tests on main and on noslp.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_icp_gpu.py -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/pytest_icp_r04i.log 2>&1; echo "pytest icp (main) exit $?"; tail -3 gpurun_out/pytest_icp_r04i.log
ER_HIP_LIB=$PWD/elasticreconstruction_amd/_ab/liber_hip_noslp.so timeout 300 python -m pytest tests/test_icp_gpu.py -q -m gpu --tb=short -p no:cacheprovider > gpurun_out/pytest_icp_r04i_noslp.log 2>&1; echo "pytest icp (noslp) exit $?"; tail -3 gpurun_out/pytest_icp_r04i_noslp.log
for rep in 1 2; do
  for v in main noslp exact exactnoslp; do
    if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$PWD/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
    echo "== $v"; ER_PROBE_SHARES="6" ER_PROBE_CLOUDS=0 timeout 300 python scripts/icp_list_probe.py 50 12 2>&1 | tail -3
  done
done
