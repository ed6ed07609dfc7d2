"""Synthetic FragmentOptimizer scenes shared by the CPU and GPU tests."""
import numpy as np
from scipy.spatial import cKDTree

from elasticreconstruction_amd import synth


def make_scene(num=3, n=20000, seed=70, res=8, length=3.0):
    """`num` fragments sampled from the synthetic room, each in its own cube frame [0, length)^3, with poses
    (fragment -> world), noisy initial poses and correspondence lists built by exact NN in the world frame
    (rows = (index in fragment i, index in fragment j), like corres_<i>_<j>.txt)."""
    rng = np.random.default_rng(seed)
    frags, poses = [], []
    base = synth.look_at((1.5, 1.5, 1.5), (0, 0, 1)) @ np.linalg.inv(synth.basepose())
    for f in range(num):
        D = synth.perturbation(seed + 10 * f, 4.0, 0.06) if f else np.eye(4)
        frag = base @ D                                   # this fragment's cube frame in the world
        x, nn = synth.sample_fragment(frag, n, seed=seed + f)
        inside = ((x > 1e-3) & (x < length - 1e-3)).all(1)
        frags.append((x[inside].astype(np.float32), nn[inside].astype(np.float32)))
        poses.append(frag)
    world = [(x.astype(np.float64) @ P[:3, :3].T + P[:3, 3]) for (x, _), P in zip(frags, poses)]
    pairs = []
    for i in range(num):
        for j in range(i + 1, num):
            d, k = cKDTree(world[i]).query(world[j])
            jj = np.nonzero(d < 0.02)[0]
            pr = np.stack([k[jj], jj], 1).astype(np.int32)
            pairs.append((i, j, pr))
    init = [P @ synth.perturbation(seed + 100 + f, 0.3, 0.004) if f else P for f, P in enumerate(poses)]
    return dict(num=num, res=res, length=length, frags=frags, poses=poses, init=init, pairs=pairs, rng=rng)


def lattice_ctr(num, res, length, poses, jitter, rng):
    """expand_ctr-like vector: every fragment's (res+1)^3 control vertices (vertex i + j*(res+1) + k*(res+1)^2, xyz
    interleaved) mapped by its pose, plus a small random deformation."""
    ul = length / res
    k, j, i = np.meshgrid(np.arange(res + 1), np.arange(res + 1), np.arange(res + 1), indexing="ij")
    v = np.stack([i.ravel() * ul, j.ravel() * ul, k.ravel() * ul], 1)
    out = []
    for P in poses:
        w = v @ P[:3, :3].T + P[:3, 3] + rng.normal(0, jitter, v.shape)
        out.append(w.reshape(-1))
    return np.concatenate(out)
