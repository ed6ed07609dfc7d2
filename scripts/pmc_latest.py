#!/usr/bin/env python3
"""profiles/pmc_latest.json from ONE scripts/pmc_summary.py summary (the file bench.py reads the static counter figures from).
usage: python scripts/pmc_latest.py profiles/<tag>_pmc_summary.txt <tag> "<what the kernel was>" [profiles/<tag>_kernel_stats.csv] > profiles/pmc_latest.json
Run it on the tree the profiled run used: the sha16 of the kernel sources is stamped into the file, and bench.py marks the static
figures 'stale' when the sources have changed since."""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def parse(path):
    k, out = None, {}
    for line in open(path):
        m = re.match(r"^(k_[a-z_0-9]+)", line)
        if m:
            k = m.group(1)
            out.setdefault(k, {})
            continue
        m = re.match(r"^\s+([A-Z_0-9]+)(?: \(under the counters\))?\s+n=\s*(\d+)\s+mean\s+([0-9.]+)", line)
        if m and k:
            out[k][m.group(1)] = float(m.group(3))
            out[k]["n"] = int(m.group(2))
            continue
        m = re.match(r"^\s+CLOCK_GHZ.*?([0-9.]+)\s*$", line)
        if m and k:
            out[k]["CLOCK_GHZ"] = float(m.group(1))
    return out


def main():
    path, tag, what = sys.argv[1], sys.argv[2], sys.argv[3]
    s = parse(path)
    ki = s["k_integrate"]
    fetch, write = ki["FETCH_SIZE"], ki["WRITE_SIZE"]                  # KiB per launch
    raw = (fetch + write) * 1024.0
    m = lambda k: s.get(k, {}).get("SQ_INSTS_VALU", 0.0) / 1e6
    import bench
    extra = {"kernel_source_sha16": bench.kernel_source_sha16(),
             "valu_wave_instructions_per_batch": {k: s[k]["SQ_INSTS_VALU"] for k in ("k_integrate", "k_reproject_scatter", "k_prepare")
                                                  if "SQ_INSTS_VALU" in s.get(k, {})}}
    if len(sys.argv) > 4:                                              # rocprofv3 --kernel-trace --stats of the same tree: AverageNs of k_integrate
        by = {}
        for row in csv.DictReader(open(sys.argv[4])):
            for kname in ("k_integrate", "k_prepare", "k_reproject_scatter"):
                if kname + "(" in row.get("Name", "") or kname + "<" in row.get("Name", ""):
                    by[kname] = float(row["AverageNs"]) / 1e3
                    if kname == "k_integrate":
                        extra["rocprof_kernel_trace_avg_us"] = by[kname]
                        extra["rocprof_kernel_trace_calls"] = int(row["Calls"])
                        extra["rocprof_kernel_trace_file"] = sys.argv[4]
        extra["rocprof_kernel_trace_avg_us_by_kernel"] = by            # the three kernels of a batch, same run (bench.py: roofline.kernels)
    print(json.dumps({**extra, **{
        "source": "%s: rocprofv3 --kernel-trace --pmc ... in separate passes (FETCH_SIZE | WRITE_SIZE | SQ_*), bench.py --steps 20 --warmup 1 "
                  "--no-alone --no-streamed: mean over the %d launches of 50 frames of a whole 3000-frame pass + warm-up; %s.  ALL figures below "
                  "come from this one run (scripts/pmc_latest.py)." % (path, ki["n"], what),
        "run": "the rocprofv3 --pmc run %s" % tag,
        "kernel": "k_integrate",
        "fetch_size_kib_per_launch": fetch, "write_size_kib_per_launch": write, "raw_bytes_per_launch": raw,
        "correction": "FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md (rocprofv3 tallies 128-B read requests at 64 B; calibrated "
                      "there for wide coalesced streaming reads) -- an UPPER bound here: the reads are 8-byte float2 rows and 4-byte gathers, for "
                      "which the counter is uncalibrated; Infinity-Cache hits (the 79 MB of scaled depth per batch) are counted too.  WRITE_SIZE as reported.",
        "k_integrate_hbm_bytes_per_launch": (2 * fetch + write) * 1024.0,
        "k_integrate_hbm_bytes_per_launch_uncorrected": raw,
        "k_integrate_valu_wave_instructions_per_launch": ki["SQ_INSTS_VALU"],
        "grbm_gui_active_per_launch_sum_over_xcds": ki.get("GRBM_GUI_ACTIVE"),
        "valu": "SQ_INSTS_VALU per 50-frame batch (same run): k_integrate %.1f M, k_reproject_scatter %.1f M, k_prepare %.1f M wave-instructions"
                % (m("k_integrate"), m("k_reproject_scatter"), m("k_prepare")),
        "measured_clock_ghz": ki.get("CLOCK_GHZ"),
        "measured_clock_source": "the same run: GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 / per-dispatch duration of k_integrate under the counters",
    }}, indent=1))


if __name__ == "__main__":
    main()
