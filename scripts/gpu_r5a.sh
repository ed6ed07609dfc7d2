#!/bin/bash
# PREPARED at the end of round 4, not run yet: three steps on the straight-line part of the NN search, compiled out by default.
#   here (build container), before the call:
#     bash scripts/build_variant.sh rowmask -DER_NN_ROWMASK=1
#     bash scripts/build_variant.sh reserve -DER_NN_ONE_RESERVE=1
#     bash scripts/build_variant.sh rcp     -DER_NN_RCP_CELL=1
#     bash scripts/build_variant.sh all3    -DER_NN_ROWMASK=1 -DER_NN_ONE_RESERVE=1 -DER_NN_RCP_CELL=1
#   then:  gpurun --timeout 700 -- bash scripts/gpu_r5a.sh
# 1. parity of every variant (the whole path-B GPU file through ER_HIP_LIB: oracle, reference CCorresApp, the adversarial margin queries),
# 2. the 50-pair list, shipped build against the variants, two interleaved rounds (three-call phases + hard list).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
AB=$PWD/elasticreconstruction_amd/_ab
VARIANTS=""
for v in rowmask reserve rcp all3; do [ -f $AB/liber_hip_$v.so ] && VARIANTS="$VARIANTS $v"; done
[ -n "$VARIANTS" ] || { echo "build the variants first (see the head of this script)"; exit 1; }
for v in $VARIANTS; do
  ER_HIP_LIB=$AB/liber_hip_$v.so timeout 300 python -m pytest tests/test_icp_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/pytest_icp_$v.log 2>&1; echo "$v: pytest exit $?"; tail -2 gpurun_out/pytest_icp_$v.log
done
for rep in 1 2; do
  for v in main $VARIANTS; do
    if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$AB/liber_hip_$v.so; fi
    echo "== $v"; ER_PROBE_FUSED=0 ER_PROBE_CLOUDS=0 timeout 300 python scripts/icp_list_probe.py 50 12 2>&1 | tail -2
  done
done
