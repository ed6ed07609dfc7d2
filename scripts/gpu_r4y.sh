#!/bin/bash
# round 4: where do the 4.9 ms of er_cloud_create_batch (25 fragments, 150 MB from page-locked memory) go?  HIP API + kernel + copy timeline of one call
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; export TMPDIR=/tmp; mkdir -p gpurun_out
python scripts/cloud_build_probe.py 25 250000 6 2>&1 | tail -1
cd /tmp && timeout 300 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_r04y -o t -- python $R/scripts/cloud_build_probe.py 25 250000 4 > $R/gpurun_out/cloud_probe_r04y.log 2>&1
cd $R; tail -1 gpurun_out/cloud_probe_r04y.log
python scripts/cloud_timeline.py /tmp/prof_r04y > gpurun_out/cloud_timeline_r04y.txt 2>&1; head -12 gpurun_out/cloud_timeline_r04y.txt | cut -c1-400
