#!/bin/bash
# Round 3, GPU call D: the multi-pair ICP (pair groups) -- parity tests, bench icp section, kernel stats.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03d; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
timeout 900 python -m pytest tests/test_icp_gpu.py tests/test_host_programs_gpu.py tests/test_fopt_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/pytest_gpu_$TAG.log; tail -25 gpurun_out/pytest_gpu_$TAG.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== t=${SECONDS}s bench"
timeout 600 python bench.py --cpu-sample 0 --no-streamed --no-alone --min-seconds 0.3 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r03d.json'))
print('frames/s', d['value'])
i=d['icp']
for k in ('pairs_per_s','mean_icp_iterations','phase_ms','timing','single_call_pairs_per_s','single_call_8_host_threads_pairs_per_s','hard_set','cpu_baseline','parity_checked','parity_checked_reference','cpu_port_pairs_per_s','cpu_port_note'):
    print(k, i.get(k))
PY
tail -3 gpurun_out/bench_$TAG.err
echo "== t=${SECONDS}s icp kernel stats"
bash scripts/gpu_icp_prof.sh 2>&1 | tail -16
echo "== done t=${SECONDS}s"
