#!/bin/bash
# Round 5, third GPU call: the new shipped build (padded grid + compaction chain without atomics + one ICP partial vector per slice) --
#   1. the whole path-B GPU file (incl. the new kinfu-like list) and the host-program tests; 2. interleaved A/B against the pad-only build of the previous
#   call (three-call flow, fused entry, hard list); 3. kernel trace of the three-call flow.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; SECONDS=0
timeout 500 python -m pytest tests/test_icp_gpu.py tests/test_host_programs_gpu.py -q -m gpu -p no:cacheprovider -s > gpurun_out/r5d_pytest_icp_host.log 2>&1; echo "pytest exit $? t=${SECONDS}s"
grep -a "kinfu-like\|passed\|failed\|Error\|assert" gpurun_out/r5d_pytest_icp_host.log | cut -c1-600 | tail -12
AB=$PWD/elasticreconstruction_amd/_ab
for rep in 1 2; do
  for v in main pad; do
    if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$AB/liber_hip_$v.so; fi
    echo "== $v"; ER_PROBE_SHARES="6" ER_PROBE_CLOUDS=0 timeout 300 python scripts/icp_list_probe.py 50 12 2>&1 | tail -3
  done
done
unset ER_HIP_LIB
echo "== t=${SECONDS}s kernel trace of the three-call flow (shipped build)"
( cd /tmp && ER_PROBE_FUSED=0 ER_PROBE_CLOUDS=0 ER_PROBE_HARD=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r5d -o icp -- python $OLDPWD/scripts/icp_list_probe.py 50 10 > $OLDPWD/gpurun_out/r5d_trace_run.log 2>&1 )
for f in $(find /tmp/prof_r5d -name "*kernel_stats*.csv"); do cp "$f" gpurun_out/r5d_icp_three_call_kernel_stats.csv; done
python scripts/kstats.py gpurun_out/r5d_icp_three_call_kernel_stats.csv | grep -E "k_count|k_icp|k_find|k_scan|k_compact|k_fitness|rocclr" | tee gpurun_out/r5d_icp_three_call_kernel_stats.txt
echo "== done t=${SECONDS}s"
