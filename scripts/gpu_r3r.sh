#!/bin/bash
# Round 3, GPU call R: (1) stream priorities of the pre-pass streams (ER_AUX_PRIO probe: 1 = lowest, -1 = highest) against the default;
# (2) the default bench line with the configs[3] / configs[4] child runs attached.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03r; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
one() {  # label, env assignment
  env $2 timeout 300 python bench.py --cpu-sample 0 --icp-pairs 0 --no-streamed --other-configs 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('%-14s value %8.0f fps  k_integrate %.3f ms  frac %.3f  alone %.3f ms' % ('$1', d['value'], r['avg_launch_ms'], r['frac'], r['kernel_alone']['avg_launch_ms']))"
}
for rep in 1 2; do
  one default ER_NOP=1
  one aux_lowest ER_AUX_PRIO=1
  one aux_highest ER_AUX_PRIO=-1
done > gpurun_out/ab_$TAG.txt 2>&1
cat gpurun_out/ab_$TAG.txt
echo "== t=${SECONDS}s default bench (with the child runs)"
timeout 900 python bench.py > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err; echo "rc $? t=${SECONDS}s"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench_default_r03r.json').readline())
print('value', d['value'], 'frac', d['roofline']['frac'], 'icp', d['icp']['pairs_per_s'])
print(json.dumps(d.get('other_configs'), indent=1)[:3000])
PY
echo "== done t=${SECONDS}s"
