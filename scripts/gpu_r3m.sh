#!/bin/bash
# Round 3, GPU call N: + interleaved target records (xyz | normal, 32 B) for the matched-point gathers.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03n; mkdir -p gpurun_out/prof_$TAG; export TMPDIR=/tmp
SECONDS=0
timeout 900 python -m pytest tests/test_icp_gpu.py tests/test_host_programs_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|error" | tail -5
echo "== t=${SECONDS}s list timing"
python scripts/icp_list_probe.py 50 20 2>&1 | tail -1
python scripts/icp_list_probe.py 50 20 2>&1 | tail -1
echo "== t=${SECONDS}s timeline"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_$TAG -o t -- python $R/scripts/icp_list_probe.py 50 6 > $R/gpurun_out/prof_$TAG/run.log 2>&1
cd $R; grep phases gpurun_out/prof_$TAG/run.log; python scripts/icp_timeline.py /tmp/prof_$TAG > gpurun_out/prof_$TAG/timeline.txt 2>&1; grep -v "copyB" gpurun_out/prof_$TAG/timeline.txt | cut -c1-120 | head -70
echo "== done t=${SECONDS}s"
