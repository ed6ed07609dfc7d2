#!/bin/bash
# Interleaved path-B A/B of library variants on ONE box:  scripts/icp_ab.sh ROUNDS name1 name2 ...   ("main" = the in-tree build)
rounds=$1; shift
for i in $(seq $rounds); do
  for v in "$@"; do
    if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$PWD/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
    python scripts/icp_ab.py 9 2>/dev/null | tail -1
  done
done
unset ER_HIP_LIB
