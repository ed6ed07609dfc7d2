#!/usr/bin/env python3
"""configs[4]'s all-pairs Registration (100 fragments, 4950 pairs): iteration histogram of the accepted pairs, phase times, and -- for the pairs that use up
many iterations -- the CPU oracle's answer and the conditioning of their point-to-plane systems.  usage: [ER_HIP_LIB=...] python scripts/icp_allpairs_probe.py"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from elasticreconstruction_amd import synth
from elasticreconstruction_amd.icp import Cloud, count_inliers_batch, icp_align_batch
n_frag = 100
frs = synth.fragment_set(n_frag, 250000, radius=0.6, device="cuda:0")
clouds = [Cloud(x, n, 0.03, 0) for x, n, _ in frs]
allp = [(i, j) for i in range(n_frag) for j in range(i + 1, n_frag)]
Ts = [np.linalg.inv(frs[i][2]) @ frs[j][2] @ synth.perturbation(9000 + i * n_frag + j, 1.0, 0.01) for i, j in allp]
for rep in range(2):
    t0 = time.perf_counter()
    cnts = count_inliers_batch([clouds[j] for _, j in allp], [clouds[i] for i, _ in allp], Ts, 0.03)
    npts = np.array([[len(clouds[i]), len(clouds[j])] for i, j in allp], np.float64)
    acc = (cnts >= 40000) | ((cnts / npts[:, 0] > 0.25) & (cnts / npts[:, 1] > 0.25))
    ai = np.nonzero(acc)[0]
    t1 = time.perf_counter()
    fins, iters, conv, _ = icp_align_batch([clouds[allp[k][1]] for k in ai], [clouds[allp[k][0]] for k in ai], [Ts[k].astype(np.float32) for k in ai], 0.03, 20, 1e-6, 0)
    t2 = time.perf_counter()
iters = np.asarray(iters)
hist = {int(v): int(c) for v, c in zip(*np.unique(iters, return_counts=True))}
out = {"lib": os.path.basename(os.environ.get("ER_HIP_LIB", "main")), "accepted": int(len(ai)), "pre_check_ms": round(1e3 * (t1 - t0), 2), "icp_ms": round(1e3 * (t2 - t1), 2),
       "iteration_histogram": hist, "mean_iterations": float(iters.mean()), "converged": int(np.sum(conv))}
long_ones = [int(ai[q]) for q in np.nonzero(iters >= 10)[0][:6]]
if os.environ.get("ER_PROBE_ORACLE", "1") == "1" and long_ones:
    from oracle.pyoracle import IcpOracle
    from oracle import refcheck
    det = {}
    for k in long_ones:
        i, j = allp[k]
        oa, ob = IcpOracle(frs[i][0], frs[i][1], 0.03), IcpOracle(frs[j][0], frs[j][1], 0.03)
        To, ito, co, _ = ob.align(oa, Ts[k].astype(np.float32), 0.03, 20, 1e-6, 0)
        q = int(np.nonzero(ai == k)[0][0])
        det[k] = {"pair": [i, j], "gpu_iterations": int(iters[q]), "oracle_iterations": int(ito), "oracle_converged": bool(co),
                  "conditioning_at_guess": refcheck.point_to_plane_conditioning_at(frs, (i, j, None), Ts[k], 0.03), "max_abs_T_diff": float(np.abs(To - fins[q]).max())}
    out["pairs_with_10_or_more_iterations"] = det
print(json.dumps(out), flush=True)
