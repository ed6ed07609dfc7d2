#!/bin/bash
# round 4, call 4: the cell-task NN search (main; ctunroll = its neighbour loop unrolled) against the round-3 row search (variant rows):
# parity of path B on main, list timing of the three builds interleaved, PMC of main.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_icp_gpu.py tests/test_host_programs_gpu.py -q -m gpu --tb=short -p no:cacheprovider -k "not fragment_optimizer and not integrate_program" > gpurun_out/pytest_icp_r04d.log 2>&1; echo "pytest icp exit $?"; tail -5 gpurun_out/pytest_icp_r04d.log
for rep in 1 2; do
  for v in main ctunroll rows; do
    if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$PWD/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
    echo "== $v"; ER_PROBE_SHARES="3" timeout 200 python scripts/icp_list_probe.py 50 12 2>&1 | tail -2
  done
done
unset ER_HIP_LIB
echo "=== ICP PMC (cell tasks)"
ER_PROBE_FUSED=0 bash scripts/gpu_icp_pmc.sh r04d_cells 50 3 2>&1 | grep -v "^   " | tail -8
python scripts/icp_pmc_derive.py gpurun_out/pmc_icp_r04d_cells/summary.txt
