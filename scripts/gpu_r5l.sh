#!/bin/bash
# Round 5, k_integrate frame loop: row terms from an LDS table (tab), classification on wave masks (masks), scalar-base gathers (saddr), inner fast loop (fast).
# Parity of path A with the variant named by $1 as THE library, then an interleaved A/B of all variants.  usage: bash scripts/gpu_r5l.sh <variant> <reps> <names...>
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; V="$1"; REPS="$2"; shift; shift; mkdir -p gpurun_out; SECONDS=0
ER_HIP_LIB=$R/elasticreconstruction_amd/_ab/liber_hip_$V.so timeout 500 python -m pytest tests/test_tsdf_gpu.py tests/test_host_programs_gpu.py -m gpu -q --tb=short -p no:cacheprovider -n 4 > gpurun_out/r05l_pytest_$V.log 2>&1
echo "pytest ($V) exit $? after ${SECONDS}s" >> gpurun_out/r05l_pytest_$V.log; tail -5 gpurun_out/r05l_pytest_$V.log
bash scripts/ab_libs.sh $REPS "$@" > gpurun_out/r05l_ab.txt 2>&1; cat gpurun_out/r05l_ab.txt
echo "== done t=${SECONDS}s"
