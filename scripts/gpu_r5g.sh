#!/bin/bash
# Round 5, sixth GPU call: (1) the tests added or changed since the rehearsal (kinfu-like list on the open-path pair list, the device-resident hand-off, the
# chain on own outputs); (2) counters for the realistic list against the uniform one: two rocprofv3 --pmc passes each (SQ set, cache set) over the three-call flow.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; SECONDS=0; R=$PWD
timeout 700 python -m pytest tests/test_icp_gpu.py tests/test_fopt_gpu.py tests/test_host_programs_gpu.py -q -m gpu -p no:cacheprovider -s --tb=short -k "kinfu or hand_off or chain" > gpurun_out/r5g_pytest.log 2>&1; echo "pytest exit $? t=${SECONDS}s"
grep -a "kinfu-like\|chain:\|passed\|failed\|Error\|assert " gpurun_out/r5g_pytest.log | cut -c1-900 | tail -14
for LIST in uniform kinfu; do
  OUT=$R/gpurun_out/pmc_icp_r5g_$LIST; mkdir -p $OUT
  [ $LIST = kinfu ] && export ER_PROBE_KINFU=1 || unset ER_PROBE_KINFU
  i=0
  for CS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    ( cd /tmp && ER_PROBE_FUSED=0 ER_PROBE_CLOUDS=0 ER_PROBE_HARD=0 timeout 300 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d /tmp/pmc_r5g_${LIST}_$i -o p$i -- python $R/scripts/icp_list_probe.py 50 3 > $OUT/run_$i.log 2>&1 )
    for f in $(find /tmp/pmc_r5g_${LIST}_$i -name "*counter_collection.csv"); do cp "$f" $OUT/pass${i}_counter_collection.csv; done
    tail -1 $OUT/run_$i.log
  done
  python scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
  echo "== $LIST (t=${SECONDS}s)"; python scripts/icp_pmc_derive.py $OUT/summary.txt 2>&1 | tee $OUT/derived.txt | grep -E "^k_|VALU instructions|L2|duration"
  rm -f $OUT/pass*_counter_collection.csv
done
echo "== done t=${SECONDS}s"
