#!/bin/bash
# Round 3, GPU call I: source pixels per thread of k_reproject_scatter (depth loads issued up front): 1 (sp1, as before), 2 (main), 4 (sp4).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03I; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
timeout 900 python -m pytest tests/test_tsdf_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -3
ER_HIP_LIB=$R/elasticreconstruction_amd/_ab/liber_hip_sp4.so timeout 600 python -m pytest tests/test_tsdf_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider -k "golden or config2 or randomised or reproject" 2>&1 | tail -2
bash scripts/ab_libs.sh 3 main sp1 sp4 > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
