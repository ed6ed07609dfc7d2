#!/bin/bash
# Round 3, GPU call E: NN search with own-cell-first scanning -- ICP parity tests + icp bench section (3 runs).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03e; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
timeout 900 python -m pytest tests/test_icp_gpu.py tests/test_host_programs_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu_$TAG.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> gpurun_out/pytest_gpu_$TAG.log; tail -8 gpurun_out/pytest_gpu_$TAG.log
for r in 1 2 3; do
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-streamed --no-alone --min-seconds 0.05 > gpurun_out/bench_${TAG}_$r.json 2> gpurun_out/bench_$TAG.err; python - <<PY
import json
d=json.loads(open('gpurun_out/bench_${TAG}_$r.json').readline())
i=d['icp']
print('pairs/s %.0f' % i['pairs_per_s'], i['phase_ms'], i['timing'], 'hard %.0f it %.1f err %.2g' % (i['hard_set']['pairs_per_s'], i['hard_set']['mean_icp_iterations'], i['hard_set']['max_abs_T_error_vs_ground_truth']), 'single %.0f thr8 %.0f' % (i['single_call_pairs_per_s'], i['single_call_8_host_threads_pairs_per_s']), i['parity_checked']['ok'], i.get('parity_checked_reference',{}).get('ok'))
PY
done
echo "== t=${SECONDS}s icp kernel stats"
bash scripts/gpu_icp_prof.sh 2>&1 | grep -v "^{" | tail -12
echo "== done t=${SECONDS}s"
