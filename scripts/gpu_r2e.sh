#!/bin/bash
# Round 2, GPU call E: full -m gpu suite (zero-crossing extraction, block-sparse Cholesky, faster device ICP step, host run-ahead),
# default bench, bench --config 4 and --config 5, kernel stats of the ICP part.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="${1:-r02e}"; mkdir -p gpurun_out
SECONDS=0
timeout 800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 4 > gpurun_out/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/pytest_gpu_$TAG.log; tail -30 gpurun_out/pytest_gpu_$TAG.log
echo "== t=${SECONDS}s bench"
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err; echo "bench exit $?"; tail -1 gpurun_out/bench_default_$TAG.json | cut -c1-400; tail -3 gpurun_out/bench_default_$TAG.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_default_$TAG.json").read().strip().splitlines()[-1])
print(json.dumps({k:d.get(k) for k in ("timing","streamed","parity_checked")}))
print(json.dumps({k:d["roofline"].get(k) for k in ("frac","avg_launch_ms","hbm_physical_frac","whole_job_frac")}))
print(json.dumps({k:d["icp"].get(k) for k in ("pairs_per_s","mean_icp_iterations","roofline","cloud_build_ms","pairs_per_s_incl_cloud_build","phase_ms","timing","single_call_pairs_per_s","parity_checked")}))
PY
echo "== t=${SECONDS}s config 4"
timeout 500 python bench.py --config 4 > gpurun_out/bench_config4_$TAG.json 2> gpurun_out/bench_config4_$TAG.err; echo "exit $?"; tail -1 gpurun_out/bench_config4_$TAG.json | cut -c1-1500; tail -3 gpurun_out/bench_config4_$TAG.err
echo "== t=${SECONDS}s config 5"
timeout 500 python bench.py --config 5 --cpu-sample 0 --no-streamed > gpurun_out/bench_config5_$TAG.json 2> gpurun_out/bench_config5_$TAG.err; echo "exit $?"; tail -1 gpurun_out/bench_config5_$TAG.json | cut -c1-600; tail -3 gpurun_out/bench_config5_$TAG.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_config5_$TAG.json").read().strip().splitlines()[-1])
    print(json.dumps(d.get("icp")))
except Exception as e: print("config5 parse", e)
PY
echo "== t=${SECONDS}s stats"
bash scripts/gpu_prof.sh $TAG --steps 20 --warmup 2 --cpu-sample 0 --icp-pairs 50 --no-streamed --min-seconds 0.2 > /dev/null 2>&1; python scripts/kstats.py gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv 2>&1 | head -24
echo "== done t=${SECONDS}s"
