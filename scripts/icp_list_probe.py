"""The bench's 50-pair list through the three batch entry points, N times, with wall-clock per phase (for API-trace profiling)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from elasticreconstruction_amd import synth
from elasticreconstruction_amd.icp import Cloud, count_inliers_batch, find_correspondence_batch, icp_align_batch
n_pairs, n_frag, reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50, 25, int(sys.argv[2]) if len(sys.argv) > 2 else 20
if os.environ.get("ER_PROBE_KINFU", "0") == "1":          # the realistic list of bench.py's icp.realistic leg (synth.kinfu_fragment, sweeps 7.2 degrees apart)
    frs = []
    for i in range(n_frag):
        x, n, F, st = synth.kinfu_fragment(i, 2 * n_frag, 250000, noise_mm=2.0 if i % 2 else 0.0)
        ok = ~np.isnan(n).any(axis=1)
        frs.append((np.ascontiguousarray(x[ok]), np.ascontiguousarray(n[ok]), F))
    clouds = [(Cloud(x, n, 0.03, 0), F) for x, n, F in frs]
    pairs = synth.chain_pair_list(frs, n_pairs, 2.0, 0.02, 700)
    os.environ.setdefault("ER_PROBE_HARD", "0")
else:
    frs = synth.fragment_set(n_frag, 250000, device="cuda:0")
    clouds = [(Cloud(x, n, 0.03, 0), F) for x, n, F in frs]
    pairs = []
    for k in range(n_pairs):
        a = k % n_frag
        b = (a + 1 + (k // n_frag) % 3) % n_frag
        pairs.append((a, b, np.linalg.inv(clouds[a][1]) @ clouds[b][1] @ synth.perturbation(700 + k, 2.0, 0.02)))
srcs, tgts = [clouds[b][0] for _, b, _ in pairs], [clouds[a][0] for a, _, _ in pairs]
Ts = [T for _, _, T in pairs]
T32 = [T.astype(np.float32) for T in Ts]
ph = []
for r in range(reps + 2):
    t0 = time.perf_counter()
    cnts = count_inliers_batch(srcs, tgts, Ts, 0.03)
    t1 = time.perf_counter()
    fins, iters, _, _ = icp_align_batch(srcs, tgts, T32, 0.03, 20, 1e-6, 0)
    t2 = time.perf_counter()
    lists, _ = find_correspondence_batch(srcs, tgts, [F.astype(np.float64) for F in fins], 0.015, 0.8660, True, copy=False)
    t3 = time.perf_counter()
    if r >= 2:
        ph.append((t1 - t0, t2 - t1, t3 - t2))
ph = np.array(ph) * 1e3
if os.environ.get("ER_PROBE_FUSED", "1") == "1":
    from elasticreconstruction_amd.icp import registration_batch
    for shares in os.environ.get("ER_PROBE_SHARES", "1 2 3 4").split():
        os.environ["ER_ICP_SHARES"] = shares
        tt = []
        for r in range(reps + 2):
            t0 = time.perf_counter()
            out = registration_batch(srcs, tgts, Ts, 0.03, 40000, 0.25, 20, 1e-6, 0, 0.015, 0.8660, want_info=True, copy=False)
            tt.append(time.perf_counter() - t0)
        tt = np.array(tt[2:]) * 1e3
        same = all(np.array_equal(out["T"][k], fins[k]) for k in range(n_pairs)) and [len(l) for l in out["lists"]] == [len(l) for l in lists]
        print("fused er_registration_batch, %s share(s): median %.2f ms (min %.2f max %.2f) -> %.0f pairs/s; accepted %d / %d; equals the three calls bit for bit: %s"
              % (shares, np.median(tt), tt.min(), tt.max(), n_pairs / np.median(tt) * 1e3, int(out["accepted"].sum()), n_pairs, same))
if os.environ.get("ER_PROBE_HARD", "1") == "1":
    hard = synth.hard_pair_list([(None, None, F) for _, F in clouds], n_pairs)
    hT = [T for _, _, T in hard]
    tt = []
    for r in range(6):
        t0 = time.perf_counter()
        count_inliers_batch(srcs, tgts, hT, 0.03)
        hf, hit, _, _ = icp_align_batch(srcs, tgts, [T.astype(np.float32) for T in hT], 0.03, 20, 1e-6, 0)
        find_correspondence_batch(srcs, tgts, [F.astype(np.float64) for F in hf], 0.015, 0.8660, True, copy=False)
        tt.append(time.perf_counter() - t0)
    tt = np.array(tt[2:]) * 1e3
    print("hard list (6 deg / 6 cm): median %.2f ms -> %.0f pairs/s; iterations mean %.2f max %d, at the limit %d"
          % (np.median(tt), n_pairs / np.median(tt) * 1e3, np.mean(hit), int(np.max(hit)), int(np.sum(np.asarray(hit) >= 20))))
if os.environ.get("ER_PROBE_CLOUDS", "1") == "1":
    from elasticreconstruction_amd import _ffi
    arena = _ffi.PinnedArena()
    arena.reset(sum(x.nbytes + n.nbytes for x, n, _ in frs) + 16384 * len(frs))
    pinned = []
    for x, n, _ in frs:
        px, pn = arena.take(x.shape, np.float32), arena.take(n.shape, np.float32)
        px[...] = x
        pn[...] = n
        pinned.append((px, pn))
    pageable = [(x, n) for x, n, _ in frs]
    for name, arrs in (("pageable", pageable), ("page-locked", pinned)):
        t1, tb = [], []
        for r in range(4):
            t0 = time.perf_counter()
            cs = [Cloud(x, n, 0.03, 0) for x, n in arrs]
            t1.append(time.perf_counter() - t0)
            [c.close() for c in cs]
            t0 = time.perf_counter()
            cs = Cloud.create_batch(arrs, 0.03, 0)
            tb.append(time.perf_counter() - t0)
            [c.close() for c in cs]
        print("cloud build, %d fragments of %d points, %s input: one by one %.2f ms, er_cloud_create_batch %.2f ms (%.0f us per fragment, %.1f GB/s of input)"
              % (len(frs), len(frs[0][0]), name, np.median(t1[1:]) * 1e3, np.median(tb[1:]) * 1e3, np.median(tb[1:]) * 1e6 / len(frs),
                 sum(x.nbytes + n.nbytes for x, n in arrs) / np.median(tb[1:]) / 1e9))
print("phases ms median", np.median(ph, 0), "min", ph.min(0), "max", ph.max(0), "total median %.2f -> %.0f pairs/s" % (np.median(ph.sum(1)), n_pairs / np.median(ph.sum(1)) * 1e3))
