#!/bin/bash
# Round 5, fifth GPU call: a rehearsal of the gate on the current head -- the whole -m gpu suite, smoke(), the default bench line -- and the merge plan probe
# (what the frame-split merge moves at 1 / 2 / 4 / 8 ranks, measured on one GPU).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; SECONDS=0
timeout 700 python -m pytest tests -q -m gpu -p no:cacheprovider -s --tb=short > gpurun_out/r5f_pytest_gpu.log 2>&1; echo "pytest exit $? t=${SECONDS}s"
grep -a "kinfu-like\|passed\|failed\|Error\|assert \|sampled stream" gpurun_out/r5f_pytest_gpu.log | cut -c1-700 | tail -14
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5f_smoke.log 2>&1; echo "smoke exit $? t=${SECONDS}s"; tail -2 gpurun_out/r5f_smoke.log
timeout 200 python scripts/merge_plan_probe.py > gpurun_out/r5f_merge_plan.txt 2>&1; echo "merge plan exit $? t=${SECONDS}s"; grep -a "configs\|export" gpurun_out/r5f_merge_plan.txt | cut -c1-900 | head -12
timeout 400 python bench.py > gpurun_out/r5f_bench_default.json 2> gpurun_out/r5f_bench_default.err; echo "bench exit $? t=${SECONDS}s"
python - <<'PY'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r5f_bench_default.json") if l.startswith("{")][-1])
    r, i = d["roofline"], d["icp"]
    print("bench: %.0f frames/s frac %.3f kernel_frac %.3f | icp %.0f pairs/s fused %.0f hard %.0f incl_build %.0f / batched %.0f" % (
        d["value"], r["frac"], r["kernel_frac"], i["pairs_per_s"], i["fused_entry"]["pairs_per_s"], i["hard_set"]["pairs_per_s"],
        i["pairs_per_s_incl_cloud_build"], i["pairs_per_s_incl_cloud_build_batched"]))
    print("realistic:", json.dumps(i.get("realistic"))[:1500])
    print("device_hand_off:", json.dumps(i.get("device_hand_off"))[:900])
    print("parity:", (d.get("parity_checked") or {}).get("bit_exact"), (i.get("parity_checked_reference") or {}).get("ok"), ((i.get("hard_set") or {}).get("parity_checked_reference") or {}).get("ok"))
    oc = d.get("other_configs") or {}
    for k in ("configs[3]", "configs[4]"):
        c = oc.get(k) or {}
        print(k, c.get("value"), (c.get("parity_checked") or {}).get("bit_exact"), c.get("error"), "wall", c.get("wall_s"))
except Exception as ex:
    print("bench: no line:", ex)
PY
tail -3 gpurun_out/r5f_bench_default.err
echo "== done t=${SECONDS}s"
