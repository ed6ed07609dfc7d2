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
