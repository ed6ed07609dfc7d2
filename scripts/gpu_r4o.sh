#!/bin/bash
# round 4, call: any-hit early exit in the Registration pre-check (main) against the full exact search (noanyhit); path-B tests on main
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_icp_gpu.py tests/test_host_programs_gpu.py -q -m gpu --tb=short -p no:cacheprovider -k "not fragment_optimizer and not integrate_program" > gpurun_out/pytest_icp_r04o.log 2>&1; echo "pytest icp exit $?"; tail -3 gpurun_out/pytest_icp_r04o.log
for rep in 1 2; do
  for v in main noanyhit; do
    if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$PWD/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
    echo "== $v"; ER_PROBE_SHARES="6" ER_PROBE_CLOUDS=0 timeout 300 python scripts/icp_list_probe.py 50 12 2>&1 | tail -3
  done
done
