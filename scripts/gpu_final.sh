#!/bin/bash
# GPU side of scripts/final_gate.sh (also usable alone for mid-round validation): the driver's gate in the driver's order --
#   1. python -m pytest tests -x -q -m gpu   (serial, -x: exactly what the driver runs; tests/conftest.py orders path A, path B, host
#      programs first and the widening rows last)
#   2. smoke()
#   3. the default bench line
# then, while the clock allows (GATE_ONLY=1 skips them): kernel-trace stats, one PMC pass (FETCH_SIZE / WRITE_SIZE / SQ_*).
# Every log starts with the HEAD scripts/final_gate.sh stamped into .gate_head ("unstamped" when run outside the gate).
#   GATE_X="" (mid-round runs): do not stop at the first failing test.   GATE_TESTS="tests/a.py tests/b.py": a subset (a change confined to one path, when
#   the round's GPU minutes no longer cover the whole suite; the log's first line then names the subset).   usage: bash scripts/gpu_final.sh <tag> [seconds]
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG="${1:-r04}"; LIMIT="${2:-600}"; mkdir -p gpurun_out
HEAD_ID="$(cat .gate_head 2>/dev/null || echo unstamped)"
SECONDS=0
LOG=gpurun_out/pytest_gpu_$TAG.log
echo "# HEAD $HEAD_ID  tag $TAG  $(date -u +%FT%TZ)  tests: ${GATE_TESTS:-tests (all)}" > $LOG
timeout $((LIMIT > 700 ? 560 : LIMIT * 4 / 5)) python -m pytest ${GATE_TESTS:-tests} ${GATE_X--x} -q -m gpu --tb=short -p no:cacheprovider >> $LOG 2>&1
PRC=$?; echo "pytest exit $PRC after ${SECONDS}s" >> $LOG; tail -6 $LOG
echo "# HEAD $HEAD_ID" > gpurun_out/smoke_$TAG.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/smoke_$TAG.log 2>&1; SRC=$?; echo "smoke exit $SRC" >> gpurun_out/smoke_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log
echo "== t=${SECONDS}s bench"
BT=$((LIMIT - SECONDS - 5)); [ $BT -gt 300 ] && BT=300; [ $BT -lt 20 ] && BT=20
if [ "${GATE_BENCH:-1}" = "1" ]; then timeout $BT python bench.py --full-json gpurun_out/bench_full_$TAG.json > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err; BRC=$?; else BRC=0; fi
python - "$TAG" "$HEAD_ID" <<'PY'
import json, sys
tag, head = sys.argv[1], sys.argv[2]
p = "gpurun_out/bench_default_%s.json" % tag
try:
    lines = [l for l in open(p).read().splitlines() if l.strip()]
    d = json.loads(lines[-1])                                  # the LAST stdout line is the compact line (< 4 KB) the driver parses
    assert len(lines[-1]) < 4096, "compact line is %d bytes" % len(lines[-1])
    d["gate_head"] = head
    open(p, "w").write(json.dumps(d, separators=(",", ":")) + "\n")
    r, i, o = d.get("roofline", {}), d.get("icp", {}), d.get("other_configs", {})
    print("bench: %.1f frames/s, frac %.3f, kernel_frac %.3f, icp %.0f / hard %.0f / kinfu-like %.0f pairs/s, configs[3] %.0f; parity %s / %s / %s; line %d bytes" % (
        d["value"], r.get("frac") or 0, r.get("kernel_frac") or 0, i.get("pairs_per_s") or 0, i.get("hard_pairs_per_s") or 0, i.get("realistic_pairs_per_s") or 0,
        (o.get("configs[3]") or {}).get("value") or 0, (d.get("parity_checked") or {}).get("bit_exact"), i.get("parity_ref_ok"), i.get("hard_ref_ok"), len(lines[-1])))
except Exception as ex:
    print("bench: no JSON line (%s)" % ex)
PY
echo "== gate: pytest $PRC smoke $SRC bench $BRC at t=${SECONDS}s (HEAD $HEAD_ID)"
if [ "${GATE_ONLY:-0}" != "1" ]; then
  echo "== t=${SECONDS}s stats"
  if [ $SECONDS -lt $((LIMIT - 90)) ]; then bash scripts/gpu_prof.sh $TAG --steps 20 --warmup 2 --cpu-sample 0 --icp-pairs 0 --other-configs 0 --boundary 0 --min-seconds 0.1 > /dev/null 2>&1; python scripts/kstats.py gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv 2>&1 | head -12; fi
  echo "== t=${SECONDS}s pmc"
  # three short passes (the MI355X guide: --pmc runs carry --kernel-trace only): SQ instruction counts + clock, FETCH_SIZE, WRITE_SIZE -- what scripts/pmc_latest.py needs
  if [ $SECONDS -lt $((LIMIT - 150)) ]; then
    OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT; i=0
    for CS in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      ( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --pmc $CS --output-format csv -d /tmp/pmc_${TAG}_$i -o p$i -- python $R/bench.py --steps 20 --warmup 1 --cpu-sample 0 --icp-pairs 0 --other-configs 0 --boundary 0 --no-streamed --no-alone --min-seconds 0.01 > $OUT/run_$i.log 2>&1 )
      for f in $(find /tmp/pmc_${TAG}_$i -name "*counter_collection.csv"); do cp "$f" $OUT/pass${i}_counter_collection.csv; done
    done
    python scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; rm -f $OUT/pass*_counter_collection.csv
    grep -A12 "^k_integrate\|^k_reproject_scatter\|^k_prepare" $OUT/summary.txt | head -60
  fi
fi
echo "== done t=${SECONDS}s"
[ $PRC -eq 0 ] && [ $SRC -eq 0 ] && [ $BRC -eq 0 ]
