#!/bin/bash
# Kernel timelines (rocprofv3 --kernel-trace) of library variants: scripts/gpu_timeline.sh TAG variant...
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="$1"; shift; mkdir -p gpurun_out/tl_$TAG
export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$R/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_${TAG}_$v -o t -- python $R/bench.py --steps 20 --warmup 1 --cpu-sample 0 --icp-pairs 0 --no-streamed --no-alone --other-configs 0 --min-seconds 0.01 > $R/gpurun_out/tl_$TAG/run_$v.log 2>&1
  cd $R
  f=$(find /tmp/tl_${TAG}_$v -name "*kernel_trace.csv" | head -1)
  cp "$f" gpurun_out/tl_$TAG/${v}_kernel_trace.csv
  echo "=== $v: $(grep -o '"value": [0-9.]*' gpurun_out/tl_$TAG/run_$v.log | head -1)"
  python scripts/timeline.py gpurun_out/tl_$TAG/${v}_kernel_trace.csv 30 3
done
