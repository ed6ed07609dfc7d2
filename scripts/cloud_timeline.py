"""Timeline of the LAST er_cloud_create_batch of scripts/cloud_build_probe.py from rocprofv3 --hip-trace --kernel-trace --memory-copy-trace CSVs
(batches are separated by 50 ms pauses): device side (kernels, copies) and the host's HIP calls.   python scripts/cloud_timeline.py <dir>"""
import csv, glob, re, sys
d = sys.argv[1]
dev, api = [], []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"::(k_[a-z_0-9]+)", n)
        n = m.group(1) if m else ("rocprim:" + ("merge" if "merge" in n else "sort" if "sort" in n else "scan" if "scan" in n else n[:12]))
        dev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "?")[-14:]))
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
dev.sort(); api.sort()
# the last group of device events that contains k_grid_cells, groups split at gaps > 20 ms
groups, cur = [], []
for e in dev:
    if cur and e[0] - max(x[1] for x in cur) > 20e6:
        groups.append(cur); cur = []
    cur.append(e)
if cur: groups.append(cur)
groups = [g for g in groups if any(e[2] in ("k_grid_cells", "k_chunk_cells") for e in g)]
g = groups[-1]
t0, t1 = min(e[0] for e in g), max(e[1] for e in g)
a = [e for e in api if e[1] >= t0 - 2e6 and e[0] <= t1 + 1e6]
ta = min([e[0] for e in a] + [t0])
print("device span %.1f us; first host call %.1f us before the first device event" % ((t1 - t0) / 1e3, (t0 - ta) / 1e3))
tot = {}
for s, e, n in g: tot[n] = tot.get(n, [0, 0.0]); tot[n][0] += 1; tot[n][1] += (e - s) / 1e3
print("device totals:", {k: (v[0], round(v[1], 1)) for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])})
cp = sorted([e for e in g if e[2].startswith("copy:") and (e[1] - e[0]) > 20e3])
if cp:
    busy = sum(e[1] - e[0] for e in cp)
    print("large copies: %d, busy %.1f us of the span %.1f us (first starts at %.1f, last ends at %.1f)" % (len(cp), busy / 1e3, (t1 - t0) / 1e3, (cp[0][0] - ta) / 1e3, (cp[-1][1] - ta) / 1e3))
    gaps = [(cp[i + 1][0] - cp[i][1]) / 1e3 for i in range(len(cp) - 1)]
    print("gaps between consecutive large copies (us):", " ".join("%.0f" % x for x in gaps))
hot = {}
for s, e, n in a: hot[n] = hot.get(n, [0, 0.0]); hot[n][0] += 1; hot[n][1] += (e - s) / 1e3
print("host calls:", {k: (v[0], round(v[1], 1)) for k, v in sorted(hot.items(), key=lambda kv: -kv[1][1])[:12]})
print("-- merged timeline (us from the first host call) --")
ev = [(s, e, "dev  " + n) for s, e, n in g] + [(s, e, "host " + n) for s, e, n in a if (e - s) > 15e3 or n in ("hipEventSynchronize", "hipStreamSynchronize")]
for s, e, n in sorted(ev)[:400]:
    print("%9.1f -> %9.1f (%7.1f)  %s" % ((s - ta) / 1e3, (e - ta) / 1e3, (e - s) / 1e3, n))
