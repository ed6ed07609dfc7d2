#!/bin/bash
# Round 3, GPU call D2: four register rows per lane again (4 x 8 x 8 patches: the longest item -- a patch that crosses the surface -- is half as long),
# with 2 / 3 / 4 persistent workgroups per CU, against the eight-row kernel (main), all with the plan records and the one-instruction round.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03D; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
ER_HIP_LIB=$R/elasticreconstruction_amd/_ab/liber_hip_r4b4.so timeout 600 python -m pytest tests/test_tsdf_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider -k "golden or config2 or randomised" 2>&1 | tail -3
AB_ALONE=1 bash scripts/ab_libs.sh 2 main r4b2 r4b3 r4b4 > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
