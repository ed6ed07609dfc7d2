#!/bin/bash
# round 4, call 3: the LDS-staged NN search (main) against the global search (variant nostage): parity of path B, list timing (three calls
# and the fused entry), PMC after; the two FragmentOptimizer failure-path tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_icp_gpu.py tests/test_host_programs_gpu.py -q -m gpu --tb=short -p no:cacheprovider -k "not fragment_optimizer" > gpurun_out/pytest_icp_r04c.log 2>&1; echo "pytest icp exit $?"; tail -5 gpurun_out/pytest_icp_r04c.log
timeout 200 python -m pytest tests/test_fopt_gpu.py -q -m gpu --tb=short -p no:cacheprovider -k "cholesky or slac_with" > gpurun_out/pytest_fopt_r04c.log 2>&1; echo "pytest fopt exit $?"; tail -5 gpurun_out/pytest_fopt_r04c.log
for rep in 1 2; do
  for v in main nostage; do
    if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$PWD/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
    echo "== $v"; ER_PROBE_SHARES="1 3" timeout 200 python scripts/icp_list_probe.py 50 12 2>&1 | tail -3
  done
done
unset ER_HIP_LIB
echo "=== ICP PMC after (staged)"
ER_PROBE_FUSED=0 bash scripts/gpu_icp_pmc.sh r04c_staged 50 3 2>&1 | tail -70
