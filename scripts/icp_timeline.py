"""Timeline of ONE pass of the 50-pair list from rocprofv3 --kernel-trace / --memory-copy-trace CSVs: python scripts/icp_timeline.py <dir>"""
import csv, glob, re, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"::(k_[a-z_0-9]+)", n)
        n = m.group(1) if m else n[:18]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, int(r.get("Grid_Size", 0) or 0)))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "?")[-12:], 0))
ev.sort()
if not any(e[2].startswith('k_count_inliers') for e in ev):
    print('files:', glob.glob(d + '/**/*.csv', recursive=True)); print('names:', sorted(set(e[2] for e in ev))[:40])
# the last k_count_inliers marks the start of the last pass
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_count_inliers")]
i0 = starts[-2] if len(starts) > 1 else starts[-1]
i1 = starts[-1]
t0 = ev[i0][0]
busy = 0
last_end = t0
for s, e, n, g in ev[i0:i1]:
    print("%-20s %9.1f -> %9.1f  (%7.1f us)  gap %7.1f  grid %d" % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - last_end) / 1e3, g))
    last_end = max(last_end, e)
print("pass length %.1f us" % ((ev[i1][0] - t0) / 1e3))
