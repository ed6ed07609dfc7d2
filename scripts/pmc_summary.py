#!/usr/bin/env python3
"""Aggregate rocprofv3 counter_collection CSVs per kernel: mean counter value per dispatch."""
import csv, glob, os, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "pass*_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "anonymous namespace)::k_" not in k:
            continue
        k = k.split("(anonymous namespace)::", 1)[1].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("   %-34s n=%4d  mean %16.1f  max %16.1f" % (c, len(v), sum(v) / len(v), max(v)))
