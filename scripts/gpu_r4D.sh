#!/bin/bash
# round 4, last profile call: rocprofv3 --kernel-trace --stats of the 50-pair list on the final build (compacted row tasks, chunked cloud build)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_icp5 -o icp -- python $R/scripts/icp_list_probe.py 50 6 > $R/gpurun_out/prof_icp_r04D.log 2>&1
cd $R; for f in $(find /tmp/prof_icp5 -name "*kernel_stats.csv"); do cp "$f" gpurun_out/r04D_icp_kernel_stats.csv; done
python scripts/kstats.py gpurun_out/r04D_icp_kernel_stats.csv | head -24
grep -v "^W2026\|rocprofv3\]" gpurun_out/prof_icp_r04D.log | tail -8
