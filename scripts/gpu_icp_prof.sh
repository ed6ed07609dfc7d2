#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/prof_icp; export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_icp -o icp -- python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --other-configs 0 --icp-pairs 40 > $R/gpurun_out/prof_icp/run.log 2>&1
cd "$R"; for f in $(find /tmp/prof_icp -name "*stats*.csv"); do cp "$f" gpurun_out/prof_icp/; done
python scripts/kstats.py gpurun_out/prof_icp/icp_kernel_stats.csv | grep -E "k_count|k_icp|k_find|k_scan|k_compact|k_init|k_fitness|rocclr"
grep '^{"metric"' gpurun_out/prof_icp/run.log | python -c "import json,sys; print(json.loads(sys.stdin.readline())['icp'])"
