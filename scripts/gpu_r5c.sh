#!/bin/bash
# Round 5, second GPU call: the padded-grid NN search (-DER_NN_PAD=1: two rings of empty cells, one 16-byte bounds load per row; pad = four rows' bounds in
# flight, pad8 = all eight) -- parity of both builds through the whole path-B GPU file, interleaved A/B on the 50-pair list, and a kernel trace of the
# three-call flow ALONE (no fused entry, no hard list) on the shipped build: where the ICP phase's 2.3 ms go.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; SECONDS=0
AB=$PWD/elasticreconstruction_amd/_ab
for v in pad pad8; do
  ER_HIP_LIB=$AB/liber_hip_$v.so timeout 300 python -m pytest tests/test_icp_gpu.py -q -m gpu -p no:cacheprovider > gpurun_out/r5c_pytest_icp_$v.log 2>&1; echo "$v: pytest exit $? t=${SECONDS}s"; tail -2 gpurun_out/r5c_pytest_icp_$v.log
done
for rep in 1 2; do
  for v in main pad pad8; do
    if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$AB/liber_hip_$v.so; fi
    echo "== $v"; ER_PROBE_FUSED=0 ER_PROBE_CLOUDS=0 timeout 300 python scripts/icp_list_probe.py 50 12 2>&1 | tail -2
  done
done
unset ER_HIP_LIB
echo "== t=${SECONDS}s kernel trace of the three-call flow (shipped build)"
( cd /tmp && ER_PROBE_FUSED=0 ER_PROBE_CLOUDS=0 ER_PROBE_HARD=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r5c -o icp -- python $OLDPWD/scripts/icp_list_probe.py 50 10 > $OLDPWD/gpurun_out/r5c_trace_run.log 2>&1 )
for f in $(find /tmp/prof_r5c -name "*kernel_stats*.csv"); do cp "$f" gpurun_out/r5c_icp_three_call_kernel_stats.csv; done
python scripts/kstats.py gpurun_out/r5c_icp_three_call_kernel_stats.csv | grep -E "k_count|k_icp|k_find|k_scan|k_compact|k_fitness|rocclr" | tee gpurun_out/r5c_icp_three_call_kernel_stats.txt
tail -1 gpurun_out/r5c_trace_run.log
echo "== done t=${SECONDS}s"
