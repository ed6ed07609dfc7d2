#!/usr/bin/env python3
"""Path B A/B on ONE box: the bench's three pair lists (uniform surfels, the 6 deg / 6 cm hard list, kinfu-like fragments) through the three batch entry
points with the library ER_HIP_LIB names (one process per variant: scripts/icp_ab.sh), median wall time per phase, and a digest of everything the calls
return -- inlier counts, iteration counts, transforms (bits), correspondence lists -- so that two variants can be compared for IDENTITY, not only speed.
usage: ER_HIP_LIB=... python scripts/icp_ab.py [reps]"""
import hashlib, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from elasticreconstruction_amd import synth
from elasticreconstruction_amd.icp import Cloud, count_inliers_batch, find_correspondence_batch, icp_align_batch

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 9
n_pairs, n_frag = 50, 25
cache = os.environ.get("ER_AB_CACHE", "/tmp/er_icp_ab_cache.npz")


def fragments():
    """both fragment sets, cached on disk so that every variant sees the same bytes (and the kinfu sweeps are integrated once)"""
    if os.path.exists(cache):
        z = np.load(cache)
        return ([(z["ux%d" % i], z["un%d" % i], z["uF%d" % i]) for i in range(n_frag)], [(z["kx%d" % i], z["kn%d" % i], z["kF%d" % i]) for i in range(n_frag)])
    uni = [(np.ascontiguousarray(x), np.ascontiguousarray(n), F) for x, n, F in synth.fragment_set(n_frag, 250000, device="cuda:0")]
    kin = []
    for i in range(n_frag):
        x, n, F, st = synth.kinfu_fragment(i, 2 * n_frag, 250000, noise_mm=2.0 if i % 2 else 0.0)
        ok = ~np.isnan(n).any(axis=1)
        kin.append((np.ascontiguousarray(x[ok]), np.ascontiguousarray(n[ok]), F))
    d = {}
    for tag, fr in (("u", uni), ("k", kin)):
        for i, (x, n, F) in enumerate(fr):
            d["%sx%d" % (tag, i)], d["%sn%d" % (tag, i)], d["%sF%d" % (tag, i)] = x, n, F
    np.savez(cache, **d)
    return uni, kin


def run(name, frs, pairs):
    clouds = [Cloud(x, n, 0.03, 0) for x, n, _ in frs]
    srcs, tgts = [clouds[b] for _, b, _ in pairs], [clouds[a] for a, _, _ in pairs]
    Ts = [T for _, _, T in pairs]
    T32 = [T.astype(np.float32) for T in Ts]
    ph, h = [], None
    for r in range(reps + 3):
        t0 = time.perf_counter()
        cnts = count_inliers_batch(srcs, tgts, Ts, 0.03)
        t1 = time.perf_counter()
        fins, iters, conv, _ = icp_align_batch(srcs, tgts, T32, 0.03, 20, 1e-6, 0)
        t2 = time.perf_counter()
        lists, infos = find_correspondence_batch(srcs, tgts, [F.astype(np.float64) for F in fins], 0.015, 0.8660, True, copy=False)
        t3 = time.perf_counter()
        if r >= 3:
            ph.append((t1 - t0, t2 - t1, t3 - t2))
        if r == 0:
            hh = hashlib.sha256()
            hh.update(np.asarray(cnts, np.int64).tobytes()); hh.update(np.asarray(iters, np.int64).tobytes()); hh.update(np.asarray(conv, np.int64).tobytes())
            hi = hh.hexdigest()[:12]
            hh.update(np.asarray(fins, np.float32).tobytes())
            hT = hh.hexdigest()[:12]
            for l in lists:
                hh.update(np.asarray(l).tobytes())
            h = {"ints": hi, "ints+T": hT, "all": hh.hexdigest()[:12], "iters_mean": float(np.mean(iters)), "corr_mean": float(np.mean([len(l) for l in lists]))}
    ph = np.array(ph) * 1e3
    tot = ph.sum(1)
    out = {"list": name, "ms": [round(float(x), 3) for x in np.median(ph, 0)], "total_ms": round(float(np.median(tot)), 3), "min_ms": round(float(tot.min()), 3),
           "pairs_per_s": round(n_pairs / float(np.median(tot)) * 1e3), "digest": h}
    for c in clouds:
        c.close()
    return out


uni, kin = fragments()
res = {"lib": os.path.basename(os.environ.get("ER_HIP_LIB", "main"))}
res["uniform"] = run("uniform", uni, synth.config2_pair_list(uni, n_pairs))
res["hard"] = run("hard", uni, synth.hard_pair_list(uni, n_pairs))
res["kinfu"] = run("kinfu", kin, synth.chain_pair_list(kin, n_pairs, 2.0, 0.02, 700))
print(json.dumps(res), flush=True)
