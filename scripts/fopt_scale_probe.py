"""Non-rigid FragmentOptimizer system kept dense in HBM: assemble + factor + solve time for `num` fragments at resolution 8
(num * 2187 unknowns).  usage: python scripts/fopt_scale_probe.py [num]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from scipy.spatial import cKDTree
from elasticreconstruction_amd import synth
from elasticreconstruction_amd.fopt import FragmentOptimizer
num = int(sys.argv[1]) if len(sys.argv) > 1 else 20
base = synth.look_at((1.5, 1.5, 1.5), (0, 0, 1)) @ np.linalg.inv(synth.basepose())
frags, poses = [], []
for f in range(num):
    P = base @ (synth.perturbation(70 + 10 * f, 4.0, 0.06) if f else np.eye(4))
    x, n = synth.sample_fragment(P, 60000, seed=70 + f)
    ok = ((x > 1e-3) & (x < 3.0 - 1e-3)).all(1)
    frags.append((x[ok].astype(np.float32), n[ok].astype(np.float32))); poses.append(P)
world = [(x.astype(np.float64) @ P[:3, :3].T + P[:3, 3]) for (x, _), P in zip(frags, poses)]
pairs = []
for i in range(num - 1):
    for j in (i + 1, i + 2):
        if j < num:
            d, k = cKDTree(world[i]).query(world[j]); jj = np.nonzero(d < 0.02)[0]
            pairs.append((i, j, np.stack([k[jj], jj], 1).astype(np.int32)))
g = FragmentOptimizer(num, 8, 3.0)
for f, (x, n) in enumerate(frags):
    assert g.SetCloud(f, x, n) == -1
ng = g.SetCorrespondences(pairs)
lat = g._canonical_lattice().reshape(-1, 3)
ctr = np.concatenate([g._apply(P, lat).reshape(-1) for P in poses])
g.UpdateAllNormal(ctr)
M = num * g.nper_
t0 = time.perf_counter(); g.FactorNonrigid(1.0); t1 = time.perf_counter()
rhs = np.random.default_rng(0).normal(size=M)
x = g.Solve(rhs); t2 = time.perf_counter()
g.FactorNonrigid(1.0); t3 = time.perf_counter()
print("%d fragments, %d pairs, %d correspondences, %d groups: system %d x %d (%.1f GB dense)" % (num, len(pairs), sum(p[2].shape[0] for p in pairs), ng, M, M, M * M * 8 / 1e9))
print("assemble + scatter + Cholesky: first %.2f s (loads rocBLAS), again %.2f s; one solve %.3f s; |x| %.3g" % (t1 - t0, t3 - t2, t2 - t1, np.abs(x).max()))
