"""er_cloud_create_batch over the bench's 25 fragments from page-locked arrays, a few times with a pause in between (for timeline profiling):
python scripts/cloud_build_probe.py [fragments] [points] [reps]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from elasticreconstruction_amd import synth, _ffi
from elasticreconstruction_amd.icp import Cloud
n_frag = int(sys.argv[1]) if len(sys.argv) > 1 else 25
n_pts = int(sys.argv[2]) if len(sys.argv) > 2 else 250000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
frs = synth.fragment_set(n_frag, n_pts, device="cuda:0")
arena = _ffi.PinnedArena()
arena.reset(sum(x.nbytes + n.nbytes for x, n, _ in frs) + 16384 * len(frs))
pinned = []
for x, n, _ in frs:
    px, pn = arena.take(x.shape, np.float32), arena.take(n.shape, np.float32)
    px[...] = x
    pn[...] = n
    pinned.append((px, pn))
resident = [Cloud(x, n, 0.03, 0) for x, n, _ in frs] if os.environ.get("ER_CBP_RESIDENT", "0") == "1" else []   # (the bench keeps 25 clouds resident)
nap = 0.0 if os.environ.get("ER_CBP_NOSLEEP", "0") == "1" else 0.05
tb = []
for r in range(reps):
    time.sleep(nap)
    t0 = time.perf_counter()
    cs = Cloud.create_batch(pinned, 0.03, 0)
    tb.append(time.perf_counter() - t0)
    time.sleep(nap)
    [c.close() for c in cs]
nbytes = sum(x.nbytes + n.nbytes for x, n in pinned)
print("resident %d, pause %.2f s:" % (len(resident), nap), "er_cloud_create_batch, %d fragments of %d points, page-locked input: %s ms -> median %.2f ms (%.0f us per fragment, %.1f GB/s of input)"
      % (n_frag, n_pts, " ".join("%.2f" % (t * 1e3) for t in tb), np.median(tb[1:]) * 1e3, np.median(tb[1:]) * 1e6 / n_frag, nbytes / np.median(tb[1:]) / 1e9))
