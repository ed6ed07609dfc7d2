#!/usr/bin/env python3
"""Aggregate rocprofv3 counter_collection CSVs per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)          # kernel -> {(file, dispatch id): duration in ns} (one entry per dispatch, whatever the counters)
for f in sorted(glob.glob(os.path.join(d, "pass*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        tag = "(anonymous namespace)::" if "anonymous namespace)::k_" in k else ("er_tsdf_k::" if "er_tsdf_k::k_" in k else None)
        if tag is None:
            continue
        k = k.split(tag, 1)[1].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            dur[k][(f, r.get("Dispatch_Id"))] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("   %-34s n=%4d  mean %16.1f  max %16.1f" % (c, len(v), sum(v) / len(v), max(v)))
    if dur[k]:
        dv = list(dur[k].values())
        mean_ns = sum(dv) / len(dv)
        print("   %-34s n=%4d  mean %16.1f  max %16.1f" % ("DURATION_NS (under the counters)", len(dv), mean_ns, max(dv)))
        g = agg[k].get("GRBM_GUI_ACTIVE")
        if g and mean_ns > 0:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles per XCD / kernel duration = engine clock while the kernel ran
            print("   %-34s         %16.3f" % ("CLOCK_GHZ = GUI_ACTIVE / 8 / duration", sum(g) / len(g) / 8.0 / mean_ns))
