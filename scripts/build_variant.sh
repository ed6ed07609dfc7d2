#!/bin/bash
# Build a variant of liber_hip.so for A/B measurements:  scripts/build_variant.sh NAME [-DFLAG ...]
# -> elasticreconstruction_amd/_ab/liber_hip_NAME.so  (select with ER_HIP_LIB=<path>)
set -e
cd "$(dirname "$0")/../elasticreconstruction_amd/csrc"
name=$1; shift
mkdir -p ../_ab/_build_$name
for f in er_common.cpp er_tsdf.hip er_icp.hip er_fopt.hip er_multi.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Wno-bitwise-instead-of-logical "$@" -I../../include -x hip -c $f -o ../_ab/_build_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../_ab/_build_$name/*.o -o ../_ab/liber_hip_$name.so
rm -rf ../_ab/_build_$name
echo built ../_ab/liber_hip_$name.so
