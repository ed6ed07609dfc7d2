#!/bin/bash
# Build liber_hip.so of a git revision as an A/B variant:  scripts/build_rev_variant.sh NAME REV [-DFLAG ...]
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"; name=$1; rev=$2; shift; shift
T=$(mktemp -d); git -C "$R" archive "$rev" elasticreconstruction_amd/csrc include | tar -x -C "$T"
mkdir -p "$R/elasticreconstruction_amd/_ab" "$T/o"
cd "$T/elasticreconstruction_amd/csrc"
DEF="-fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-memory-clause"      # the Makefile's per-file flags of the two kernel translation units
for f in er_common.cpp er_tsdf.hip er_tsdf_pre.hip er_tsdf_int.hip er_icp.hip er_fopt.hip er_multi.hip; do
  [ -f $f ] || continue
  extra=""; [ $f = er_tsdf_pre.hip ] && extra="$DEF"; [ $f = er_tsdf_int.hip ] && extra="$DEF"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -Wno-bitwise-instead-of-logical "$@" $extra -I../../include -x hip -c $f -o "$T/o/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$T"/o/*.o -o "$R/elasticreconstruction_amd/_ab/liber_hip_$name.so" -ldl
rm -rf "$T"; echo "built _ab/liber_hip_$name.so from $rev"
