#!/bin/bash
# PREPARED at the end of round 4, not run yet: the per-cell row mask of the NN search, -DER_NN_ROWMASK=1.
#   here (build container):   bash scripts/build_variant.sh rowmask -DER_NN_ROWMASK=1
#   then:                     gpurun --timeout 600 -- bash scripts/gpu_r5a.sh
# 1. parity of the variant (the whole path-B GPU file through ER_HIP_LIB), 2. the 50-pair list, shipped build against the variant, two interleaved rounds.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
V=$PWD/elasticreconstruction_amd/_ab/liber_hip_rowmask.so
[ -f "$V" ] || { echo "build the variant first: bash scripts/build_variant.sh rowmask -DER_NN_ROWMASK=1"; exit 1; }
ER_HIP_LIB=$V timeout 400 python -m pytest tests/test_icp_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/pytest_icp_rowmask.log 2>&1; echo "variant pytest exit $?"; tail -3 gpurun_out/pytest_icp_rowmask.log
for rep in 1 2; do
  for v in main rowmask; do
    if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$V; fi
    echo "== $v"; ER_PROBE_FUSED=0 ER_PROBE_CLOUDS=0 timeout 300 python scripts/icp_list_probe.py 50 12 2>&1 | tail -2
  done
done
