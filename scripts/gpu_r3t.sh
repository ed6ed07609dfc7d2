#!/bin/bash
# Round 3, GPU call T: TSDFVolume::round as ONE instruction (v_cvt_rpi_i32_f32) in pixel_index: exhaustive arithmetic check on the
# device, the path A parity suite, then A/B against the five-instruction form (pxold).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03t; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
timeout 1200 python -m pytest tests/test_tsdf_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -15
echo "== t=${SECONDS}s A/B"
bash scripts/ab_libs.sh 3 main pxold > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== done t=${SECONDS}s"
