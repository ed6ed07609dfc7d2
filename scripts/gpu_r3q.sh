#!/bin/bash
# Round 3, GPU call Q: pipeline depth 4 with 3 or 2 pre-pass streams against the default (depth 3, 2 streams), with the 8-row voxel pass.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03q; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
ER_HIP_LIB=$R/elasticreconstruction_amd/_ab/liber_hip_d43.so timeout 600 python -m pytest tests/test_tsdf_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider -k "golden or config2 or randomised or batch_boundary" 2>&1 | grep -E "passed|failed" | tail -2
echo "== t=${SECONDS}s A/B"
bash scripts/ab_libs.sh 3 main d43 d42 > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
