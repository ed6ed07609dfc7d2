#!/bin/bash
# round 4: chunked grid build of er_cloud_create_batch (one set of launches per chunk of clouds, all uploads queued up front, two copy streams)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_icp_gpu.py tests/test_host_programs_gpu.py -q -m gpu -p no:cacheprovider -x > gpurun_out/pytest_r04z.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/pytest_r04z.log
python scripts/cloud_build_probe.py 25 250000 8 2>&1 | tail -1
ER_PROBE_FUSED=0 ER_PROBE_HARD=0 timeout 300 python scripts/icp_list_probe.py 50 8 2>&1 | tail -3
cd /tmp && timeout 300 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_r04z -o t -- python $R/scripts/cloud_build_probe.py 25 250000 4 > $R/gpurun_out/cloud_probe_r04z.log 2>&1
cd $R; tail -1 gpurun_out/cloud_probe_r04z.log
python scripts/cloud_timeline.py /tmp/prof_r04z > gpurun_out/cloud_timeline_r04z.txt 2>&1; head -6 gpurun_out/cloud_timeline_r04z.txt | cut -c1-600
