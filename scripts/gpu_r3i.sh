#!/bin/bash
# Round 3, GPU call I: k_integrate with EIGHT register rows per lane (8 x 8 x 8 box per wave, half the items).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03i; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
ER_HIP_LIB=$R/elasticreconstruction_amd/_ab/liber_hip_rows8a.so timeout 600 python -m pytest tests/test_tsdf_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider -k "golden or config2 or randomised or batch_boundary" 2>&1 | tail -4
echo "== t=${SECONDS}s A/B"
bash scripts/ab_libs.sh 2 main rows8a rows8b rows8c > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
