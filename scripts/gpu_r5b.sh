#!/bin/bash
# Round 5, first GPU call: (1) the two new sampled-stream parity tests of configs[3] / configs[4] (tests/test_tsdf_gpu.py), (2) the new bench line
# (roofline.frac = whole job, per-kernel split) on the headline and on the config-4 child with its sampled parity leg, (3) scripts/gpu_r5a.sh = parity +
# interleaved A/B of the NN variants prepared at the end of round 4 (ROWMASK / ONE_RESERVE / RCP_CELL).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; SECONDS=0
timeout 300 python -m pytest tests/test_tsdf_gpu.py -q -m gpu -p no:cacheprovider -k "config4 or config5" -s > gpurun_out/r5b_pytest_configs34.log 2>&1; echo "configs34 pytest exit $? t=${SECONDS}s"; grep -a "sampled stream\|passed\|failed\|Error" gpurun_out/r5b_pytest_configs34.log | tail -6
timeout 200 python bench.py --config 4 --min-seconds 0.2 --cpu-sample 100 --no-alone --no-streamed --other-configs 0 > gpurun_out/r5b_bench_config4.json 2> gpurun_out/r5b_bench_config4.err; echo "bench config4 exit $? t=${SECONDS}s"
timeout 200 python bench.py --icp-pairs 0 --other-configs 0 --no-streamed > gpurun_out/r5b_bench_headline.json 2> gpurun_out/r5b_bench_headline.err; echo "bench headline exit $? t=${SECONDS}s"
python - <<'PY'
import json
for f in ("r5b_bench_config4", "r5b_bench_headline"):
    try:
        d = json.loads([l for l in open("gpurun_out/%s.json" % f) if l.startswith("{")][-1])
        r = d["roofline"]
        print(f, "%.0f frames/s frac %.3f kernel_frac %.3f rocprof %s" % (d["value"], r["frac"], r["kernel_frac"], r.get("kernel_frac_rocprof")),
              {k: round(v.get("frac_rocprof") or 0, 3) for k, v in r["kernels"].items()}, "parity", (d.get("parity_checked") or {}).get("bit_exact"),
              {k: (d.get("parity_checked") or {}).get(k) for k in ("units_gpu", "units_at_negative_coordinates", "frames")}, "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as ex:
        print(f, "no line:", ex)
PY
echo "== t=${SECONDS}s r5a"
bash scripts/gpu_r5a.sh 2>&1 | tee gpurun_out/r5b_ab_nn_variants.txt
echo "== done t=${SECONDS}s"
