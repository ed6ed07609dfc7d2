"""Print a per-kernel timeline (start / end in microseconds relative to the first kernel of the window) from a rocprofv3
kernel-trace CSV: python scripts/timeline.py <kernel_trace.csv> [first_integrate_index] [count]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(name):
    # "(anonymous namespace)::k_x(args)" and "void (anonymous namespace)::k_x<true>(args)" -> k_x
    if name.startswith("(anonymous namespace)::") or name.startswith("void (anonymous namespace)::"):
        name = name.split("(anonymous namespace)::", 1)[1]
    return name.split("(")[0].split("<")[0][:22]
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows]
ks.sort()
ints = [i for i, k in enumerate(ks) if k[2].startswith("k_integrate")]
first = ints[int(sys.argv[2]) if len(sys.argv) > 2 else len(ints) // 2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
lo = ks[first][0]
hi = ks[ints[ints.index(first) + n]][1] if ints.index(first) + n < len(ints) else ks[-1][1]
for s, e, name, q in ks:
    if s >= lo - 600000 and e <= hi and name.startswith("k_"):
        print("%-22s q%-3s %9.1f -> %9.1f  (%7.1f us)" % (name, q, (s - lo) / 1e3, (e - lo) / 1e3, (e - s) / 1e3))
