#!/usr/bin/env python3
"""Print our kernels from a rocprofv3 *_kernel_stats.csv (usage: python scripts/kstats.py <csv>)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r['Name']
    if 'anonymous namespace)::k_' in n or 'er_tsdf_k::k_' in n or 'rocclr' in n:
        short = n
        for tag in ('(anonymous namespace)::', 'er_tsdf_k::'):
            if tag + 'k_' in n:
                short = n.split(tag, 1)[1].split('(')[0]
        print("%-28s calls %5s avg %10.1f us  total %9.2f ms  min %9.1f max %9.1f" % (
            short[:28], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
