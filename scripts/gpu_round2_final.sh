#!/bin/bash
# Round validation (rounds 2 and 3): the WHOLE -m gpu suite sequentially (as the driver runs it), smoke, default bench line, kernel-trace stats over a
# whole 3000-frame pass, the PMC passes (separate runs, kernel-trace only), ICP kernel stats.  usage: bash scripts/gpu_round2_final.sh TAG
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="${1:-r02z}"; mkdir -p gpurun_out
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -x -q --tb=short -p no:cacheprovider --durations=10 > gpurun_out/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/pytest_gpu_$TAG.log; tail -22 gpurun_out/pytest_gpu_$TAG.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke_$TAG.log; tail -3 gpurun_out/smoke_$TAG.log
echo "== t=${SECONDS}s default bench"
timeout 600 python bench.py > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err; head -c 1200 gpurun_out/bench_default_$TAG.json; echo
echo "== t=${SECONDS}s stats (same command as the PMC passes: one whole 3000-frame pass + warm-up, no 'alone' pass)"
bash scripts/gpu_prof.sh $TAG --steps 20 --warmup 1 --cpu-sample 0 --icp-pairs 0 --no-streamed --no-alone --other-configs 0 --min-seconds 0.01 > /dev/null 2>&1; python scripts/kstats.py gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv 2>&1 | head -8
grep -o '"avg_launch_ms": [0-9.]*' gpurun_out/prof_$TAG/bench.json | head -1
echo "== t=${SECONDS}s pmc"
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for CS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d /tmp/pmc_${TAG}_$i -o p$i -- python $R/bench.py --steps 20 --warmup 1 --cpu-sample 0 --icp-pairs 0 --no-streamed --no-alone --other-configs 0 --min-seconds 0.01 > $OUT/run_$i.log 2>&1
  for f in $(find /tmp/pmc_${TAG}_$i -name "*counter_collection.csv"); do cp "$f" $OUT/pass${i}_counter_collection.csv; done
  echo "pass $i done t=${SECONDS}s"
done
cd $R; python scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; grep -A16 "^k_integrate" $OUT/summary.txt | head -18
rm -f $OUT/pass*_counter_collection.csv
echo "== t=${SECONDS}s icp kernel stats"
bash scripts/gpu_icp_prof.sh 2>&1 | tail -14
echo "== done t=${SECONDS}s"
