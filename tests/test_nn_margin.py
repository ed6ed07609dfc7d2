"""The pruning margin of the GPU nearest-neighbour search (csrc/er_icp.hip: grid_slack), checked on the CPU: the adversarial queries of
tests/nn_margin_cases.py defeat the constant 1e-12 of rounds 1-4 in float32 arithmetic, the margin sized from the grid's extent covers them, and the
oracle (27 cells, no pruning) names the point behind the face as their nearest neighbour.  tests/test_icp_gpu.py runs the same clouds through the HIP path."""
import numpy as np

from nn_margin_cases import build, grid_slack
from oracle.pyoracle import IcpOracle

f32 = np.float32


def test_adversarial_queries_defeat_the_old_margin_and_not_the_new_one():
    tgt, src, expect, rec, (org, cell, dim) = build()
    assert len(src) == 40 and dim[0] >= 100
    slack = grid_slack(dim, cell)
    assert 1e-8 < float(slack) < 1e-6                             # (a few tenths of a millimetre, squared)
    for lhs, d, dstar in rec:
        assert dstar < d                                           # the point behind the face IS closer than the competitor in the own cell ...
        assert f32(lhs) > f32(f32(f32(d) * f32(1.0001)) + f32(1e-12))      # ... the old bound skips its cell ...
        assert f32(lhs) <= f32(f32(f32(d) * f32(1.0001)) + slack)          # ... the new one scans it
    nt = np.tile(np.array([[0, 0, 1]], np.float32), (len(tgt), 1))
    ns = np.tile(np.array([[0, 0, 1]], np.float32), (len(src), 1))
    pairs, _ = IcpOracle(src, ns, 0.03).find_correspondence(IcpOracle(tgt, nt, 0.03), np.eye(4), 0.015, 0.8660)
    assert pairs.shape == (40, 2) and np.array_equal(pairs[:, 1], np.arange(40)) and np.array_equal(pairs[:, 0], expect)


def _kernel_skips(q, p, B, org, cell, slack):
    """nn_block's pruning rule, restated in float32 for a query q and a target point p that the grid build put into a NEIGHBOUR cell of q's cell:
    True where the search would not look at p's cell when its best squared distance so far is B (arrays of float32, one row per sample)."""
    u = ((q - org).astype(f32) / cell).astype(f32)
    c = np.floor(u)
    lo = ((u - c).astype(f32) * cell).astype(f32)
    hi = (cell - lo).astype(f32)
    cp = np.floor(((p - org).astype(f32) / cell).astype(f32))
    d = (cp - c).astype(np.int64)                                  # (-1, 0, 1) per axis
    assert (np.abs(d) <= 1).all() and (d != 0).any(1).all()
    e = np.where(d < 0, lo, np.where(d > 0, hi, f32(0))).astype(f32)
    bound = ((B * f32(1.0001)).astype(f32) + f32(slack)).astype(f32)
    e2 = ((e[:, 1] * e[:, 1]).astype(f32) + (e[:, 2] * e[:, 2]).astype(f32)).astype(f32)
    row_skipped = (d[:, 1:] != 0).any(1) & (e2 > bound)
    x_term = np.where((d[:, 1:] != 0).any(1), ((e[:, 0] * e[:, 0]).astype(f32) + e2).astype(f32), (e[:, 0] * e[:, 0]).astype(f32))
    return row_skipped | ((d[:, 0] != 0) & (x_term > bound))


def _dist(q, p):
    dx, dy, dz = ((q - p).astype(f32)).T
    return (((dx * dx).astype(f32) + (dy * dy).astype(f32)).astype(f32) + (dz * dz).astype(f32)).astype(f32)


def test_pruning_rule_never_hides_a_closer_point_faces_edges_corners():
    """Property behind grid_slack, fuzzed where it is tight: the query close to one, two or three faces of its cell, the target point just behind
    them (up to micrometres), the best distance so far ONE float above the point's own distance -- the rule must still look at the point's cell.
    With the constant 1e-12 of rounds 1-4 the same samples produce violations (the fuzz is sharp)."""
    rng = np.random.default_rng(11)
    cell = f32(f32(0.03) * f32(1.001))
    org = np.array([-1.7, -0.5, -0.5], np.float32)
    dim = [104, 34, 34]
    slack = grid_slack(dim, cell)
    n = 400000
    bad_old = 0
    for axes in (1, 2, 3):
        k = np.stack([rng.integers(50, 100, n), rng.integers(5, 30, n), rng.integers(5, 30, n)], 1).astype(np.float64)
        side = rng.integers(0, 2, (n, 3)) * 2 - 1                                   # which face of the cell, per axis
        near = np.zeros((n, 3), bool)
        near[:, :axes] = True
        near = rng.permuted(near, axis=1)
        depth = 10.0 ** rng.uniform(-5.0, -2.2, (n, 3))                            # query 10 um .. 6 mm inside its cell, from the chosen faces
        inside = np.where(near, depth, rng.uniform(0.008, 0.02, (n, 3)))
        face = org.astype(np.float64) + (k + (side > 0)) * float(cell)             # the chosen face plane per axis
        q = (face - side * inside).astype(np.float32)
        behind = 10.0 ** rng.uniform(-8.0, -5.5, (n, 3))                           # the target point 10 nm .. 3 um behind those faces, elsewhere next to q
        p = np.where(near, face + side * behind, q.astype(np.float64) + rng.uniform(-2e-5, 2e-5, (n, 3))).astype(np.float32)
        cq = np.floor(((q - org).astype(f32) / cell).astype(f32))
        cp = np.floor(((p - org).astype(f32) / cell).astype(f32))
        ok = (cq == k).all(1) & (np.abs(cp - cq) <= 1).all(1) & (cp != cq).any(1)  # p really was binned into a neighbour cell of q's cell
        q, p = q[ok], p[ok]
        assert len(q) > n // 4
        B = np.nextafter(_dist(q, p), f32(np.inf))                                 # a competitor that is farther by one float
        assert not _kernel_skips(q, p, B, org, cell, slack).any()
        bad_old += int(_kernel_skips(q, p, B, org, cell, 1e-12).sum())
    assert bad_old > 100
