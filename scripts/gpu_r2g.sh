#!/bin/bash
# Round 2, GPU call G: free-space path of k_integrate -- parity tests, A/B against -DER_NO_FREE_PATH, then the PMC passes
# (VALU instructions; FETCH_SIZE; WRITE_SIZE -- separate runs, kernel-trace only) and kernel-trace stats over a WHOLE 3000-frame pass.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="${1:-r02g}"; mkdir -p gpurun_out
SECONDS=0
timeout 800 python -m pytest tests/test_tsdf_gpu.py tests/test_host_programs_gpu.py -m gpu -q --tb=short -p no:cacheprovider -n 4 > gpurun_out/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/pytest_gpu_$TAG.log; tail -20 gpurun_out/pytest_gpu_$TAG.log
echo "== t=${SECONDS}s A/B"
bash scripts/ab_libs.sh 3 main nofree > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== t=${SECONDS}s stats"
bash scripts/gpu_prof.sh $TAG --steps 20 --warmup 1 --cpu-sample 0 --icp-pairs 0 --no-streamed --min-seconds 0.01 > /dev/null 2>&1; python scripts/kstats.py gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv 2>&1 | head -8
echo "== t=${SECONDS}s pmc"
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
i=0
for CS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d /tmp/pmc_${TAG}_$i -o p$i -- python $R/bench.py --steps 20 --warmup 1 --cpu-sample 0 --icp-pairs 0 --no-streamed --min-seconds 0.01 > $OUT/run_$i.log 2>&1
  for f in $(find /tmp/pmc_${TAG}_$i -name "*counter_collection.csv"); do cp "$f" $OUT/pass${i}_counter_collection.csv; done
  echo "pass $i done t=${SECONDS}s"
done
cd $R; python scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; grep -A9 "^k_integrate\|^k_reproject_scatter\|^k_prepare" $OUT/summary.txt | head -50
echo "== done t=${SECONDS}s"
