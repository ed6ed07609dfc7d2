#!/bin/bash
# Round 3, GPU call F: where does the host time of the ICP batch entry points go?  HIP API trace + kernel trace of the 50-pair list;
# plus the marching-cubes tests.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03f; mkdir -p gpurun_out/prof_$TAG; export TMPDIR=/tmp
SECONDS=0
python scripts/icp_list_probe.py 50 20 2>&1 | tail -2
echo "== t=${SECONDS}s hip trace"
cd /tmp && timeout 600 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_$TAG -o t -- python $R/scripts/icp_list_probe.py 50 20 > $R/gpurun_out/prof_$TAG/run.log 2>&1
cd $R; for f in $(find /tmp/prof_$TAG -name "*stats*.csv"); do cp "$f" gpurun_out/prof_$TAG/; done
tail -2 gpurun_out/prof_$TAG/run.log
ls gpurun_out/prof_$TAG
for f in gpurun_out/prof_$TAG/*hip_api_stats.csv gpurun_out/prof_$TAG/*hip*stats.csv; do [ -f "$f" ] && head -25 "$f"; done | cut -c1-200
python scripts/kstats.py gpurun_out/prof_$TAG/t_kernel_stats.csv 2>&1 | head -14
echo "== t=${SECONDS}s mesh tests"
timeout 600 python -m pytest tests/test_tsdf_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider -k "mesh or zero_crossing" 2>&1 | tail -15
echo "== done t=${SECONDS}s"
