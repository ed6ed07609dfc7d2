#!/bin/bash
# Round 5, fourth GPU call: (1) path-B GPU file incl. the kinfu-like list, the frame-split / merge tests of path A and the host programs on the build
# with per-workgroup ICP partials restored; (2) ICP chunk length 3 / 6 (rounds 2-4) against 4 / 8, same library, interleaved; (3) config-4 child of the
# bench (sparse merge on one rank: nothing moves) .
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; SECONDS=0
timeout 600 python -m pytest tests/test_icp_gpu.py tests/test_host_programs_gpu.py tests/test_tsdf_gpu.py -q -m gpu -p no:cacheprovider -s -k "not exhaustive and not full_config2 and not randomised" > gpurun_out/r5e_pytest.log 2>&1; echo "pytest exit $? t=${SECONDS}s"
grep -a "kinfu-like\|passed\|failed\|Error\|assert" gpurun_out/r5e_pytest.log | cut -c1-700 | tail -12
for rep in 1 2; do
  for ch in 3 4; do
    echo "== chunk $ch"; ER_ICP_CHUNK=$ch ER_PROBE_SHARES="6" ER_PROBE_CLOUDS=0 timeout 300 python scripts/icp_list_probe.py 50 12 2>&1 | tail -3
  done
done
echo "== t=${SECONDS}s bench config 4"
timeout 200 python bench.py --config 4 --min-seconds 0.2 --cpu-sample 100 --no-alone --no-streamed --other-configs 0 > gpurun_out/r5e_bench_config4.json 2> gpurun_out/r5e_bench_config4.err; echo "bench config4 exit $? t=${SECONDS}s"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5e_bench_config4.json") if l.startswith("{")][-1])
print("config4: %.0f frames/s, merge_stats %s, parity %s" % (d["value"], d["config"].get("merge_stats"), (d.get("parity_checked") or {}).get("bit_exact")))
PY
echo "== done t=${SECONDS}s"
