#!/usr/bin/env python3
"""SIMT cost model of the path-B NN search (design study, CPU only): how many candidate evaluations does a WAVE execute per query lane
under the two-phase search of nn_block (own cell, left / right cell of the home row in-thread; the surviving neighbour ROWS as
workgroup-shared tasks), broken down by step, and under variants of the task list: empty ranges dropped ("compact": ships since round 4),
two / three bins by length, sorted, split into chunks, the home row's side cells as tasks too.  A wave executes, for every step, the
longest trip count among its lanes: the model counts those trips (4 candidates each).
What the model is NOT (measured twice, profiles/r04d_*, r04v_*, r04w_*): a predictor of kernel time -- more than half of the instructions
of these kernels are the straight-line part every query runs, which no scan order touches; it over-predicted both experiments by ~2x.
usage: python scripts/icp_simt_sim.py [points per fragment]"""
import sys
import numpy as np
from scipy.spatial import cKDTree
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from elasticreconstruction_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
frs = synth.fragment_set(25, N)
cell = np.float32(0.03 * 1.001); R2 = 0.03**2
def grid(x):
    lo = x.min(0); dim = np.floor((x.max(0) - lo) / cell).astype(int) + 1
    q = np.clip(np.floor((x - lo) / cell).astype(int), 0, dim - 1)
    c = (q[:, 2] * dim[1] + q[:, 1]) * dim[0] + q[:, 0]
    order = np.argsort(c, kind="stable"); cs = np.zeros(dim.prod() + 1, np.int64); np.add.at(cs, c + 1, 1)
    return lo, dim, order, np.cumsum(cs), c
def trips(n, u=4): return (np.asarray(n) + u-1) // u
for (a, b, rot) in [(0, 1, 2.0), (0, 1, 1.0), (0, 1, 0.3), (0, 1, 0.0)]:
    T = np.linalg.inv(frs[a][2]) @ frs[b][2] @ synth.perturbation(700, rot, rot / 100)
    lo, dim, order, cs, tcell = grid(frs[a][0]); tx = frs[a][0]
    src = frs[b][0][grid(frs[b][0])[2]]
    q = (src @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    u = (q - lo) / cell; ic = np.floor(u).astype(int); ok = ((ic >= 0) & (ic < dim)).all(1); f = (u - ic) * cell
    cnt = lambda x, y, z: np.where((x >= 0) & (x < dim[0]) & (y >= 0) & (y < dim[1]) & (z >= 0) & (z < dim[2]),
                                   cs[np.clip((z * dim[1] + y) * dim[0] + x + 1, 0, len(cs) - 1)] - cs[np.clip((z * dim[1] + y) * dim[0] + x, 0, len(cs) - 1)], 0)
    tree = cKDTree(tx); dd, ii = tree.query(q, k=24, distance_upper_bound=0.03 * 1.8); d2 = dd.astype(np.float64) ** 2
    nb_cell = np.where(np.isfinite(dd), tcell[np.clip(ii, 0, len(tx) - 1)], -1)
    own_id = (ic[:, 2] * dim[1] + ic[:, 1]) * dim[0] + ic[:, 0]
    best_own = np.where(nb_cell == own_id[:, None], d2, np.inf).min(1); bound0 = np.minimum(R2, best_own)
    xlo, xhi = f[:, 0], cell - f[:, 0]
    need_l = (xlo ** 2 <= bound0) & (ic[:, 0] - 1 >= 0); best_l = np.where(nb_cell == (own_id - 1)[:, None], d2, np.inf).min(1)
    bound1 = np.where(need_l, np.minimum(bound0, best_l), bound0)
    need_r = (xhi ** 2 <= bound1) & (ic[:, 0] + 1 < dim[0]); best_r = np.where(nb_cell == (own_id + 1)[:, None], d2, np.inf).min(1)
    bound = np.where(need_r, np.minimum(bound1, best_r), bound1)
    c_own, c_l, c_r = cnt(ic[:, 0], ic[:, 1], ic[:, 2]), cnt(ic[:, 0] - 1, ic[:, 1], ic[:, 2]), cnt(ic[:, 0] + 1, ic[:, 1], ic[:, 2])
    ylo, yhi, zlo, zhi = f[:, 1], cell - f[:, 1], f[:, 2], cell - f[:, 2]
    n = len(q); rt_cnt = np.zeros((n, 8), np.int64); rt0 = np.zeros((n, 8), np.int64); k8=0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            if dy==0 and dz==0: continue
            ey = np.where(dy < 0, ylo, np.where(dy > 0, yhi, 0.0)); ez = np.where(dz < 0, zlo, np.where(dz > 0, zhi, 0.0)); e2 = ey ** 2 + ez ** 2
            y, z = ic[:, 1] + dy, ic[:, 2] + dz
            cx = [cnt(ic[:, 0] + dx, y, z) for dx in (-1, 0, 1)]
            live = e2 <= bound
            tot = cx[1] + np.where(xlo ** 2 + e2 <= bound, cx[0], 0) + np.where(xhi ** 2 + e2 <= bound, cx[2], 0)
            rt_cnt[:, k8] = np.where(live, tot, -1)
            live0 = e2 <= bound0
            tot0 = cx[1] + np.where(xlo ** 2 + e2 <= bound0, cx[0], 0) + np.where(xhi ** 2 + e2 <= bound0, cx[2], 0)
            rt0[:, k8] = np.where(live0, tot0, -1); k8 += 1
    S = dict(own=0., l=0., r=0., tasks=0., ntasks=0., useful=0., nohit=0.)
    V = {}
    nwg=0
    for s in range(0, n - 255, 256 * 8):
        sl = slice(s, s + 256)
        if not ok[sl].all(): continue
        nwg += 1
        for w in range(4):
            ws = slice(s + 64 * w, s + 64 * w + 64)
            S['own'] += trips(c_own[ws]).max(); S['l'] += trips(np.where(need_l[ws], c_l[ws], 0)).max(); S['r'] += trips(np.where(need_r[ws], c_r[ws], 0)).max()
        tasks = rt_cnt[sl][rt_cnt[sl] >= 0]
        S['ntasks'] += len(tasks)
        for w0 in range(0, len(tasks), 64): S['tasks'] += trips(tasks[w0:w0 + 64]).max()
        allc = np.concatenate([c_own[sl], np.where(need_l[sl], c_l[sl], 0), np.where(need_r[sl], c_r[sl], 0), tasks])
        S['useful'] += allc.sum()/4/64
        S['nohit'] += (bound[sl] >= R2).sum()
        own_t = sum(trips(c_own[s + 64 * w:s + 64 * w + 64]).max() for w in range(4))
        lr_t = sum(trips(np.where(need_l[s+64*w:s+64*w+64], c_l[s+64*w:s+64*w+64], 0)).max() + trips(np.where(need_r[s+64*w:s+64*w+64], c_r[s+64*w:s+64*w+64], 0)).max() for w in range(4))
        ne = tasks[tasks > 0]
        def waves(lst):
            return sum(trips(lst[w0:w0+64]).max() for w0 in range(0, len(lst), 64))
        def add(k, t, nt): V.setdefault(k, [0., 0.]); V[k][0] += t; V[k][1] += nt
        add('compact', own_t + lr_t + waves(ne), len(ne))
        for cut in (4, 8, 12, 16):
            add('2bins@%d' % cut, own_t + lr_t + waves(ne[ne <= cut]) + waves(ne[ne > cut]), len(ne))
        add('3bins@4,12', own_t + lr_t + waves(ne[ne <= 4]) + waves(ne[(ne > 4) & (ne <= 12)]) + waves(ne[ne > 12]), len(ne))
        add('sorted', own_t + lr_t + waves(np.sort(ne)), len(ne))
        for K in (8, 16):
            ch = []
            for c in ne:
                r = c
                while r > 0: ch.append(min(r, K)); r -= K
            add('split%d' % K, own_t + lr_t + waves(np.array(ch)), len(ch))
        side = np.concatenate([c_l[sl][need_l[sl] & (c_l[sl] > 0)], c_r[sl][need_r[sl] & (c_r[sl] > 0)]])
        alln = np.concatenate([side, ne])
        add('compact+sides', own_t + waves(alln), len(alln))
        need_r0 = (xhi ** 2 <= bound0) & (ic[:, 0] + 1 < dim[0])
        side0 = np.concatenate([c_l[sl][need_l[sl] & (c_l[sl] > 0)], c_r[sl][need_r0[sl] & (c_r[sl] > 0)]])
        t0 = rt0[sl][rt0[sl] > 0]
        all0 = np.concatenate([side0, t0])
        add('compact+sides(bound0)', own_t + waves(all0), len(all0))
        add('2bins@8+sides(bound0)', own_t + waves(all0[all0 <= 8]) + waves(all0[all0 > 8]), len(all0))
        add('2bins@8+sides', own_t + waves(alln[alln <= 8]) + waves(alln[alln > 8]), len(alln))
    per = lambda t: t*4.0/(nwg*4)
    print("pair (%d, %d), guess %.1f deg off: %d workgroups sampled; queries without a target point within the radius %.2f" % (a, b, rot, nwg, S['nohit'] / nwg / 256))
    print("   round-3 search, candidate slots a wave executes per lane:", {k: round(float(per(v)), 1) for k, v in S.items() if k not in ('ntasks', 'nohit')},
          "| (query, row) tasks per query %.2f" % (S['ntasks'] / nwg / 256))
    print("   variants of the task list (slots per lane, tasks per query):", {K: (round(float(per(v[0])), 1), round(v[1] / nwg / 256, 2)) for K, v in V.items()})
