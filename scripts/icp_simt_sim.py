#!/usr/bin/env python3
"""SIMT cost model of the path-B NN search (design study, CPU only): how many candidate evaluations does a WAVE execute per query
under (a) the shipped two-phase search of nn_block (own cell, left / right cell of the home row in-thread; the surviving neighbour ROWS
as workgroup-shared tasks) and (b) variants that compact at a finer grain (left / right cells as tasks too; one task per NON-EMPTY
cell).  A wave executes, for every step, the longest trip count among its lanes: the model counts those trips (4 candidates each).
usage: python scripts/icp_simt_sim.py [points per fragment]"""
import sys
import numpy as np
from scipy.spatial import cKDTree
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from elasticreconstruction_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 250000
frs = synth.fragment_set(25, N)
cell = np.float32(0.03 * 1.001)
R2 = 0.03 ** 2


def grid(x):
    lo = x.min(0)
    dim = np.floor((x.max(0) - lo) / cell).astype(int) + 1
    q = np.clip(np.floor((x - lo) / cell).astype(int), 0, dim - 1)
    c = (q[:, 2] * dim[1] + q[:, 1]) * dim[0] + q[:, 0]
    order = np.argsort(c, kind="stable")
    cs = np.zeros(dim.prod() + 1, np.int64)
    np.add.at(cs, c + 1, 1)
    return lo, dim, order, np.cumsum(cs), c


def trips(n):
    return (np.asarray(n) + 3) // 4


for (a, b, rot) in [(0, 1, 2.0), (0, 2, 2.0), (0, 1, 6.0)]:
    T = np.linalg.inv(frs[a][2]) @ frs[b][2] @ synth.perturbation(700, rot, rot / 100)
    lo, dim, order, cs, tcell = grid(frs[a][0])
    tx = frs[a][0]
    src = frs[b][0][grid(frs[b][0])[2]]
    q = (src @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    u = (q - lo) / cell
    ic = np.floor(u).astype(int)
    ok = ((ic >= 0) & (ic < dim)).all(1)                       # (queries one cell outside the grid are ignored by the model)
    f = (u - ic) * cell                                        # distance to the lower faces
    cnt = lambda x, y, z: np.where((x >= 0) & (x < dim[0]) & (y >= 0) & (y < dim[1]) & (z >= 0) & (z < dim[2]),
                                   cs[np.clip((z * dim[1] + y) * dim[0] + x + 1, 0, len(cs) - 1)] - cs[np.clip((z * dim[1] + y) * dim[0] + x, 0, len(cs) - 1)], 0)
    tree = cKDTree(tx)
    dd, ii = tree.query(q, k=24, distance_upper_bound=0.03 * 1.8)
    d2 = dd.astype(np.float64) ** 2
    # best squared distance inside the own cell / the home row's side cells (from the 24 nearest: exact whenever it matters)
    nb_cell = np.where(np.isfinite(dd), tcell[np.clip(ii, 0, len(tx) - 1)], -1)
    own_id = (ic[:, 2] * dim[1] + ic[:, 1]) * dim[0] + ic[:, 0]
    best_own = np.where(nb_cell == own_id[:, None], d2, np.inf).min(1)
    bound0 = np.minimum(R2, best_own)
    xlo, xhi = f[:, 0], cell - f[:, 0]
    need_l = (xlo ** 2 <= bound0) & (ic[:, 0] - 1 >= 0)
    best_l = np.where(nb_cell == (own_id - 1)[:, None], d2, np.inf).min(1)
    bound1 = np.where(need_l, np.minimum(bound0, best_l), bound0)
    need_r = (xhi ** 2 <= bound1) & (ic[:, 0] + 1 < dim[0])
    best_r = np.where(nb_cell == (own_id + 1)[:, None], d2, np.inf).min(1)
    bound = np.where(need_r, np.minimum(bound1, best_r), bound1)
    c_own, c_l, c_r = cnt(ic[:, 0], ic[:, 1], ic[:, 2]), cnt(ic[:, 0] - 1, ic[:, 1], ic[:, 2]), cnt(ic[:, 0] + 1, ic[:, 1], ic[:, 2])
    # neighbour rows / cells that survive the pruning with `bound` (shipped) or with bound0 (variants that push right after the own cell)
    rows_ship, cells_fine, cells_fine_nonempty = [], [], []
    ylo, yhi, zlo, zhi = f[:, 1], cell - f[:, 1], f[:, 2], cell - f[:, 2]
    row_tasks = []     # per query: list of candidate counts of its row tasks (shipped)
    cell_tasks0 = []   # variant: every other cell as its own task, pruned with bound0, empty cells dropped
    n = len(q)
    rt_cnt = np.zeros((n, 8), np.int64)
    ct_cnt = np.zeros((n, 26), np.int64)
    k8 = k26 = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            ey = np.where(dy < 0, ylo, np.where(dy > 0, yhi, 0.0))
            ez = np.where(dz < 0, zlo, np.where(dz > 0, zhi, 0.0))
            e2 = ey ** 2 + ez ** 2
            y, z = ic[:, 1] + dy, ic[:, 2] + dz
            cx = [cnt(ic[:, 0] + dx, y, z) for dx in (-1, 0, 1)]
            ex = [xlo ** 2, 0.0 * xlo, xhi ** 2]
            if not (dy == 0 and dz == 0):
                live = e2 <= bound
                tot = cx[1] + np.where(xlo ** 2 + e2 <= bound, cx[0], 0) + np.where(xhi ** 2 + e2 <= bound, cx[2], 0)
                rt_cnt[:, k8] = np.where(live, tot, -1)
                k8 += 1
            for j, dx in enumerate((-1, 0, 1)):
                if dx == 0 and dy == 0 and dz == 0:
                    continue
                live = (ex[j] + e2 <= bound0) & (cx[j] > 0)
                ct_cnt[:, k26] = np.where(live, cx[j], -1)
                k26 += 1
    okq = np.nonzero(ok)[0]
    tot_ship = tot_lr = tot_cell = 0.0
    useful = 0.0
    nwg = 0
    for s in range(0, n - 255, 256 * 8):                       # every 8th workgroup
        sl = slice(s, s + 256)
        if not ok[sl].all():
            continue
        nwg += 1
        # ---- shipped ----
        t = 0
        for w in range(4):
            ws = slice(s + 64 * w, s + 64 * w + 64)
            t += trips(c_own[ws]).max()
            t += trips(np.where(need_l[ws], c_l[ws], 0)).max() + trips(np.where(need_r[ws], c_r[ws], 0)).max()
        tasks = rt_cnt[sl][rt_cnt[sl] >= 0]
        for r0 in range(0, len(tasks), 256):
            rr = tasks[r0:r0 + 256]
            for w0 in range(0, len(rr), 64):
                t += trips(rr[w0:w0 + 64]).max()
        tot_ship += t
        useful += (c_own[sl].sum() + np.where(need_l[sl], c_l[sl], 0).sum() + np.where(need_r[sl], c_r[sl], 0).sum() + tasks.sum()) / 4.0 / 64.0
        # ---- variant 1: left / right cells of the home row become (single-cell) tasks of a separate list ----
        t = sum(trips(c_own[s + 64 * w:s + 64 * w + 64]).max() for w in range(4))
        side = np.concatenate([c_l[sl][need_l[sl] & (c_l[sl] > 0)], c_r[sl][need_r[sl] & (c_r[sl] > 0)]])
        for lst in (side, tasks):
            for r0 in range(0, len(lst), 256):
                rr = lst[r0:r0 + 256]
                for w0 in range(0, len(rr), 64):
                    t += trips(rr[w0:w0 + 64]).max()
        tot_lr += t
        # ---- variant 2: one task per non-empty cell that survives the bound of the own cell, tasks sorted by size within the workgroup ----
        t = sum(trips(c_own[s + 64 * w:s + 64 * w + 64]).max() for w in range(4))
        ct = np.sort(ct_cnt[sl][ct_cnt[sl] > 0])[::-1]
        for w0 in range(0, len(ct), 64):
            t += trips(ct[w0:w0 + 64]).max()
        tot_cell += t
    per_q = lambda tot: tot * 4.0 / (nwg * 4)                  # candidate evaluations a wave executes per query LANE... per wave-step x 4 candidates / 4 waves
    print((a, b, rot), "workgroups sampled", nwg,
          "| executed candidate slots per lane: shipped %.1f, sides-as-tasks %.1f, cell tasks (sorted) %.1f | useful (no SIMT waste) %.1f"
          % (per_q(tot_ship), per_q(tot_lr), per_q(tot_cell), useful * 4.0 * 64 / (nwg * 256) * 1.0))
