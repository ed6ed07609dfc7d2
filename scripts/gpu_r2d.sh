#!/bin/bash
# Round 2, GPU call D: full -m gpu suite (device-resident ICP loop, GPU grid build, LDS-staged lattice in the exact Reproject),
# default bench, A/B of the LDS variants, ICP kernel stats.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="${1:-r02d}"; mkdir -p gpurun_out
SECONDS=0
timeout 700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 4 > gpurun_out/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/pytest_gpu_$TAG.log; tail -25 gpurun_out/pytest_gpu_$TAG.log
echo "== t=${SECONDS}s bench"
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err; echo "bench exit $?"; tail -1 gpurun_out/bench_default_$TAG.json | cut -c1-600; tail -3 gpurun_out/bench_default_$TAG.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_default_$TAG.json").read().strip().splitlines()[-1])
print(json.dumps({k:d.get(k) for k in ("streamed","parity_checked")}))
print(json.dumps({k:d["icp"].get(k) for k in ("pairs_per_s","mean_icp_iterations","roofline","cloud_build_ms","pairs_per_s_incl_cloud_build","phase_ms","timing","single_call_pairs_per_s","parity_checked")}))
PY
echo "== t=${SECONDS}s A/B"
bash scripts/ab_libs.sh 2 main rs1 rs4 nolds > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== t=${SECONDS}s stats"
bash scripts/gpu_prof.sh $TAG --steps 20 --warmup 2 --cpu-sample 0 --icp-pairs 50 --no-streamed --min-seconds 0.2 > /dev/null 2>&1; python scripts/kstats.py gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv 2>&1 | head -40
echo "== done t=${SECONDS}s"
