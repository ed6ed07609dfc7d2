#!/bin/bash
# Build an A/B variant of liber_hip.so from a PATCHED COPY of the sources (the tree itself stays clean):
#   scripts/build_patched_variant.sh NAME patch.py|patch.diff [-DFLAG ...]
# patch.py is run with the copy's csrc/ as its working directory; a .diff (git diff of csrc/er_*.hip, paths a/elasticreconstruction_amd/csrc/...) is applied
# with patch -p3 there.  -> elasticreconstruction_amd/_ab/liber_hip_NAME.so
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; patch=$(realpath $2); shift; shift
T=$(mktemp -d)
mkdir -p $T/elasticreconstruction_amd $T/include
cp -r $R/elasticreconstruction_amd/csrc $T/elasticreconstruction_amd/csrc
cp $R/include/*.h $T/include/
cd $T/elasticreconstruction_amd/csrc
case "$patch" in *.diff) patch -p3 < $patch;; *) python $patch;; esac
mkdir -p $R/elasticreconstruction_amd/_ab $T/o
DEF="-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-memory-clause"
for f in er_common.cpp er_tsdf.hip er_tsdf_pre.hip er_tsdf_int.hip er_icp.hip er_fopt.hip er_multi.hip; do
  extra=""; [ $f = er_tsdf_pre.hip ] && extra="$DEF"; [ $f = er_tsdf_int.hip ] && extra="$DEF"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Wno-bitwise-instead-of-logical "$@" $extra -x hip -c $f -o $T/o/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $T/o/*.o -o $R/elasticreconstruction_amd/_ab/liber_hip_$name.so -ldl
rm -rf $T
echo built elasticreconstruction_amd/_ab/liber_hip_$name.so
