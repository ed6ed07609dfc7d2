#!/bin/bash
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03l; mkdir -p gpurun_out/prof_$TAG; export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_$TAG -o t -- python $R/scripts/icp_list_probe.py 50 6 > $R/gpurun_out/prof_$TAG/run.log 2>&1
cd $R; grep phases gpurun_out/prof_$TAG/run.log; find /tmp/prof_$TAG -name "*.csv" | head; python scripts/icp_timeline.py /tmp/prof_$TAG > gpurun_out/prof_$TAG/timeline.txt 2>&1; cat gpurun_out/prof_$TAG/timeline.txt | cut -c1-120 | head -150
