#!/usr/bin/env python3
"""Diagnostic: the ICP trajectory of ONE kinfu-like pair iteration by iteration -- the HIP path, the restatement (oracle/icp_oracle.cpp) and the reference-side
ICP of oracle/stub_corres -- by running each with max_iter = 1 .. 20 from the same guess.  usage: icp_trace_probe.py <a> <b> <list index> [rot trans seed0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from elasticreconstruction_amd import synth
from elasticreconstruction_amd.icp import Cloud, icp_align, count_inliers
from oracle.pyoracle import IcpOracle, RefCorres
a, b, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rot, trans, seed0 = (float(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (6.0, 0.06, 1700)
frs = {}
for i in (a, b):
    x, n, F, st = synth.kinfu_fragment(i, 50, 250000, noise_mm=2.0 if i % 2 else 0.0)
    ok = ~np.isnan(n).any(axis=1)
    frs[i] = (np.ascontiguousarray(x[ok]), np.ascontiguousarray(n[ok]), F)
gc = {i: Cloud(frs[i][0], frs[i][1], 0.03) for i in (a, b)}
oc = {i: IcpOracle(frs[i][0], frs[i][1], 0.03) for i in (a, b)}
gt = np.linalg.inv(frs[a][2]) @ frs[b][2]
Tg = (gt @ synth.perturbation(seed0 + k, rot, trans))
print("pre-check gpu %d oracle %d" % (count_inliers(gc[b], gc[a], Tg, 0.03), oc[b].count_inliers(oc[a], Tg, 0.03)))
Tg = Tg.astype(np.float32)
for m in range(1, 21):
    G, ig, cg, _ = icp_align(gc[b], gc[a], Tg, 0.03, m, 1e-6, 0)
    O, io, co, _ = oc[b].align(oc[a], Tg, 0.03, m, 1e-6, 0)
    S, i_s, cs, _ = RefCorres.icp(frs[b][0], frs[b][1], frs[a][0], frs[a][1], Tg, 0.03, m, 1e-6)
    d = lambda X, Y: float(np.abs(X.astype(np.float64) - Y.astype(np.float64)).max())
    print("max_iter %2d: iterations gpu %2d oracle %2d stub %2d | |gpu - oracle| %.3g  |oracle - stub| %.3g  |gpu - stub| %.3g | gt error gpu %.4f oracle %.4f stub %.4f"
          % (m, ig, io, i_s, d(G, O), d(O, S), d(G, S), d(G, gt), d(O, gt), d(S, gt)), flush=True)
