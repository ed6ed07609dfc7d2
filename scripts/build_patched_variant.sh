#!/bin/bash
# Build an A/B variant of liber_hip.so from a PATCHED COPY of the sources (the tree itself stays clean):
#   scripts/build_patched_variant.sh NAME patch.py [-DFLAG ...]
# patch.py is run with the copy's csrc/ as its working directory.  -> elasticreconstruction_amd/_ab/liber_hip_NAME.so
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; patch=$(realpath $2); shift; shift
T=$(mktemp -d)
mkdir -p $T/elasticreconstruction_amd $T/include
cp -r $R/elasticreconstruction_amd/csrc $T/elasticreconstruction_amd/csrc
cp $R/include/*.h $T/include/
cd $T/elasticreconstruction_amd/csrc
python $patch
mkdir -p $R/elasticreconstruction_amd/_ab $T/o
for f in er_common.cpp er_tsdf.hip er_icp.hip er_fopt.hip er_multi.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Wno-bitwise-instead-of-logical "$@" -x hip -c $f -o $T/o/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $T/o/*.o -o $R/elasticreconstruction_amd/_ab/liber_hip_$name.so -ldl
rm -rf $T
echo built elasticreconstruction_amd/_ab/liber_hip_$name.so
