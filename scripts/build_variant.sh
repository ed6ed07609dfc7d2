#!/bin/bash
# Build a variant of liber_hip.so for A/B measurements:  scripts/build_variant.sh NAME [-DFLAG ...]
# -> elasticreconstruction_amd/_ab/liber_hip_NAME.so  (select with ER_HIP_LIB=<path>)
# FLAGS_PRE / FLAGS_INT (environment): extra flags for er_tsdf_pre.hip (k_reproject_scatter, k_prepare) / er_tsdf_int.hip (k_integrate) only;
# unset = the Makefile's defaults for those two files, "none" = no extra flags.
set -e
cd "$(dirname "$0")/../elasticreconstruction_amd/csrc"
name=$1; shift
mkdir -p ../_ab/_build_$name
DEF="-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-memory-clause"
PRE="${FLAGS_PRE-$DEF}"; INT="${FLAGS_INT-$DEF}"; [ "$PRE" = none ] && PRE=""; [ "$INT" = none ] && INT=""
for f in er_common.cpp er_tsdf.hip er_tsdf_pre.hip er_tsdf_int.hip er_icp.hip er_fopt.hip er_multi.hip; do
  extra=""; [ $f = er_tsdf_pre.hip ] && extra="$PRE"; [ $f = er_tsdf_int.hip ] && extra="$INT"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Wno-bitwise-instead-of-logical "$@" $extra -I../../include -x hip -c $f -o ../_ab/_build_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../_ab/_build_$name/*.o -o ../_ab/liber_hip_$name.so
rm -rf ../_ab/_build_$name
echo built ../_ab/liber_hip_$name.so
