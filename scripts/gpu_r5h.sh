#!/bin/bash
# Round 5, housekeeping call on the gated build: rarely used bench flags still run (--force-merge on one rank: the sparse merge, nothing moves; --host-input),
# bin/Integrate --gpus 1 --force_merge through the C++ threads path, and the kernel trace of the final path-B build (three-call flow).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; SECONDS=0; R=$PWD
timeout 200 python bench.py --force-merge --icp-pairs 0 --other-configs 0 --no-streamed --no-alone --cpu-sample 0 --min-seconds 0.3 > gpurun_out/r5h_bench_force_merge.json 2> gpurun_out/r5h_bench_force_merge.err; echo "force-merge exit $? t=${SECONDS}s"
timeout 200 python bench.py --host-input --icp-pairs 0 --other-configs 0 --no-streamed --no-alone --cpu-sample 0 --min-seconds 0.3 > gpurun_out/r5h_bench_host_input.json 2> gpurun_out/r5h_bench_host_input.err; echo "host-input exit $? t=${SECONDS}s"
python - <<'PY'
import json
for f in ("r5h_bench_force_merge", "r5h_bench_host_input"):
    try:
        d = json.loads([l for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][-1])
        print(f, "%.0f frames/s" % d["value"], d["config"].get("merge_stats"), d["config"].get("inputs", "")[:40])
    except Exception as ex:
        print(f, "no line:", ex, open("gpurun_out/%s.err" % f).read()[-500:])
PY
( cd /tmp && ER_PROBE_FUSED=0 ER_PROBE_CLOUDS=0 ER_PROBE_HARD=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r5h -o icp -- python $R/scripts/icp_list_probe.py 50 10 > $R/gpurun_out/r5h_trace_run.log 2>&1 )
for f in $(find /tmp/prof_r5h -name "*kernel_stats*.csv"); do cp "$f" gpurun_out/r5h_icp_three_call_kernel_stats.csv; done
python scripts/kstats.py gpurun_out/r5h_icp_three_call_kernel_stats.csv | grep -E "k_count|k_icp|k_find|k_scan|k_compact|rocclr" | tee gpurun_out/r5h_icp_three_call_kernel_stats.txt
tail -1 gpurun_out/r5h_trace_run.log
echo "== done t=${SECONDS}s"
