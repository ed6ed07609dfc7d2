#!/bin/bash
# Round 3, GPU call K: parity + default bench of the 8-row k_integrate; timeline of one ICP list pass.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03k; mkdir -p gpurun_out/prof_$TAG; export TMPDIR=/tmp
SECONDS=0
timeout 900 python -m pytest tests/test_tsdf_gpu.py tests/test_host_programs_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | tail -3
echo "== t=${SECONDS}s bench"
timeout 600 python bench.py --icp-pairs 0 > gpurun_out/bench_$TAG.json 2>/dev/null; python -c "
import json; d=json.loads(open('gpurun_out/bench_r03k.json').readline()); r=d['roofline']
print(d['value'], d['ms_per_step'], r['frac'], r['avg_launch_ms'], r['kernel_alone']['avg_launch_ms'], d['parity_checked']['bit_exact'], d['streamed']['value'])"
echo "== t=${SECONDS}s icp timeline"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_$TAG -o t -- python $R/scripts/icp_list_probe.py 50 6 > $R/gpurun_out/prof_$TAG/run.log 2>&1
cd $R; tail -1 gpurun_out/prof_$TAG/run.log; python scripts/icp_timeline.py /tmp/prof_$TAG > gpurun_out/prof_$TAG/timeline.txt 2>&1; cat gpurun_out/prof_$TAG/timeline.txt | cut -c1-120 | head -150
echo "== done t=${SECONDS}s"
