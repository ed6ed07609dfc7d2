#!/bin/bash
# Round 3, GPU call M: k_icp_iter seeds the NN search of iterations >= 1 with the previous iteration's nearest neighbour (main) against the plain search (base).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03M; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
timeout 900 python -m pytest tests/test_icp_gpu.py tests/test_host_programs_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|error" | tail -5
echo "== t=${SECONDS}s list timing (main, base, main, base)"
{ for i in 1 2; do
  echo -n "main  "; python scripts/icp_list_probe.py 50 20 2>&1 | tail -1
  echo -n "base  "; ER_HIP_LIB=$R/elasticreconstruction_amd/_ab/liber_hip_base.so python scripts/icp_list_probe.py 50 20 2>&1 | tail -1
done; } | tee gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
