#!/bin/bash
# rocprofv3 kernel-trace stats of a bench.py run; CSV summaries land in gpurun_out/prof_<tag>/.
# usage: bash scripts/gpu_prof.sh <tag> [bench args...]
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="$1"; shift
mkdir -p gpurun_out/prof_$TAG
export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python $R/bench.py "$@" > $R/gpurun_out/prof_$TAG/run.log 2>&1
cd "$R"
for f in $(find /tmp/prof_$TAG -name "*stats*.csv"); do cp "$f" gpurun_out/prof_$TAG/; done
grep '^{"metric"' gpurun_out/prof_$TAG/run.log | tail -1 > gpurun_out/prof_$TAG/bench.json
ls -la gpurun_out/prof_$TAG; head -30 gpurun_out/prof_$TAG/*kernel_stats.csv
