#!/bin/bash
# One-call validation sized for a short GPU budget: -m gpu suite (4 xdist workers), smoke, default bench, one interleaved
# A/B pass over the library variants in elasticreconstruction_amd/_ab, kernel-trace stats, one PMC pass (VALU instructions).
# Later steps are skipped when the clock runs out.   usage: bash scripts/gpu_final.sh <tag> [seconds]
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="${1:-r01u}"; LIMIT="${2:-360}"; mkdir -p gpurun_out
SECONDS=0
timeout 280 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 4 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/pytest_gpu.log; tail -12 gpurun_out/pytest_gpu.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
echo "== t=${SECONDS}s bench"
BT=$((LIMIT - SECONDS - 5)); [ $BT -gt 200 ] && BT=200; [ $BT -lt 20 ] && BT=20
timeout $BT python bench.py > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err; tail -1 gpurun_out/bench_default_$TAG.json | cut -c1-700
echo "== t=${SECONDS}s A/B"
if [ $SECONDS -lt $((LIMIT - 110)) ]; then bash scripts/ab_libs.sh ${AB_REPS:-1} ${AB_LIST:-main r01s} > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt; fi
echo "== t=${SECONDS}s stats"
if [ $SECONDS -lt $((LIMIT - 60)) ]; then bash scripts/gpu_prof.sh $TAG --steps 20 --warmup 2 --cpu-sample 0 --icp-pairs 0 > /dev/null 2>&1; python scripts/kstats.py gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv 2>&1 | head -12; fi
echo "== t=${SECONDS}s pmc"
if [ $SECONDS -lt $((LIMIT - 30)) ]; then
  OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
  timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pmc_${TAG}_1 -o p1 -- python $R/bench.py --steps 6 --warmup 1 --cpu-sample 0 --icp-pairs 0 > $OUT/run_1.log 2>&1
  for f in $(find /tmp/pmc_${TAG}_1 -name "*counter_collection.csv"); do cp "$f" $OUT/pass1_counter_collection.csv; done
  cd $R; python scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; grep -A8 "^k_integrate" $OUT/summary.txt
fi
echo "== done t=${SECONDS}s"
