"""SURVEY.md 8f-2, CPU side: oracle/fopt_oracle.cpp (plain C++ restatement of FragmentOptimizer's point updates and
Hessian assembly) pinned against oracle/_ref/libref_fopt.so = the reference's own PointCloud.h compiled in place plus
OptApp.cpp's bucket expressions on the vendored Eigen.  Float32 point state must match bit for bit; the float64 bucket
values to 1e-14 relative (Eigen's fixed-size dot products add in another order); the dense assembly is checked
against an independent numpy construction from the buckets."""
import numpy as np
import pytest

from oracle.pyoracle import FoptOracle, RefFopt
from fopt_helpers import lattice_ctr, make_scene


def _load(sc, cls):
    o = cls(sc["num"], sc["res"], sc["length"])
    for f, (x, n) in enumerate(sc["frags"]):
        assert o.set_cloud(f, x, n) == -1
    return o


@pytest.mark.skipif(not RefFopt.available(), reason="oracle/_ref/libref_fopt.so needs /root/reference at build time")
def test_point_state_bitwise_and_buckets_against_reference_header():
    sc = make_scene(num=3, n=6000)
    own, ref = _load(sc, FoptOracle), _load(sc, RefFopt)

    def same_state():
        for f in range(sc["num"]):
            a, b = own.points(f), ref.points(f)
            for key in ("idx0", "val", "nval", "p", "n"):
                assert np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)), (f, key)
    same_state()                                               # GetCoordinate, PointCloud.h:92-176
    for f in range(sc["num"]):                                 # UpdatePose, :71-83
        M = sc["init"][f].astype(np.float32)
        own.update_pose(f, M)
        ref.update_pose(f, M)
    same_state()
    Rt = np.stack([P[:3, :3].T.reshape(9) for P in sc["init"]])
    rng = np.random.default_rng(5)
    for i, j, pr in sc["pairs"]:
        for k in rng.choice(pr.shape[0], 200, replace=False):
            va, ba = own.rigid_bucket(i, int(pr[k, 0]), j, int(pr[k, 1]))
            vb, bb = ref.rigid_bucket(i, int(pr[k, 0]), j, int(pr[k, 1]))
            assert np.allclose(va, vb, rtol=1e-14, atol=1e-15) and abs(ba - bb) <= 1e-15 + 1e-14 * abs(bb)
            ia, va, ba = own.slac_bucket(i, int(pr[k, 0]), j, int(pr[k, 1]), Rt)
            ib, vb, bb = ref.slac_bucket(i, int(pr[k, 0]), j, int(pr[k, 1]), Rt)
            assert np.array_equal(ia, ib) and np.allclose(va, vb, rtol=1e-14, atol=1e-15) and abs(ba - bb) <= 1e-15 + 1e-14 * abs(bb)
    ctr = lattice_ctr(sc["num"], sc["res"], sc["length"], sc["init"], 0.003, np.random.default_rng(9))   # UpdateAllPointPN, :44-52
    for f in range(sc["num"]):
        own.update_point_pn(f, ctr[f * own.nper:(f + 1) * own.nper])
        ref.update_point_pn(f, ctr)
    same_state()
    # out-of-bound point: both stop at the same index (PointCloud.cpp:57-60)
    x, n = sc["frags"][0]
    xb = x.copy()
    xb[17, 2] = 3.5
    assert own.set_cloud(0, xb, n) == 17 and ref.set_cloud(0, xb, n) == 17


def test_dense_assembly_equals_bucket_sums():
    sc = make_scene(num=3, n=3000)
    o = _load(sc, FoptOracle)
    for f in range(sc["num"]):
        o.update_pose(f, sc["init"][f].astype(np.float32))
    o.set_pairs(sc["pairs"])
    JJ, Jb, score = o.assemble_rigid()
    N = 6 * sc["num"]
    J2, b2, s2 = np.zeros((N, N)), np.zeros(N), 0.0
    for i, j, pr in sc["pairs"]:
        J2[np.arange(6), np.arange(6)] += 1.0                   # OptApp.cpp:322-324
        idx = np.r_[i * 6 + np.arange(6), j * 6 + np.arange(6)]
        for a, c in pr:
            v, b = o.rigid_bucket(i, int(a), j, int(c))
            J2[np.ix_(idx, idx)] += np.outer(v, v)
            b2[idx] += v * b
            s2 += b * b
    assert np.allclose(JJ, J2, rtol=1e-11, atol=1e-12) and np.allclose(Jb, b2, rtol=1e-11, atol=1e-12) and score == pytest.approx(s2, rel=1e-12)
    assert np.allclose(JJ, JJ.T)
    Rt = np.stack([P[:3, :3].T.reshape(9) for P in sc["init"]])
    JJ, Jb, score = o.assemble_slac(Rt)
    N = 6 * sc["num"] + o.nper
    J2, b2 = np.zeros((N, N)), np.zeros(N)
    for i, j, pr in sc["pairs"]:
        for a, c in pr[:400]:
            pass
    # full check on a sub-list (the python loop is slow): re-assemble with only 300 correspondences per pair
    sub = [(i, j, pr[:300]) for i, j, pr in sc["pairs"]]
    o.set_pairs(sub)
    JJ, Jb, score = o.assemble_slac(Rt)
    s2 = 0.0
    for i, j, pr in sub:
        for a, c in pr:
            idx, v, b = o.slac_bucket(i, int(a), j, int(c), Rt)
            S = np.zeros(N)
            np.add.at(S, idx, v)                                # duplicates add: (v_a + v_c)^2 = v_a^2 + v_c^2 + 2 v_a v_c (:539-541)
            nz = np.unique(idx)
            J2[np.ix_(nz, nz)] += np.outer(S[nz], S[nz])
            b2[idx] += 0.0
            np.add.at(b2, idx, b * v)
            s2 += b * b
    assert np.allclose(JJ, np.triu(J2), rtol=1e-11, atol=1e-13) and np.allclose(Jb, b2, rtol=1e-11, atol=1e-13)
    assert score == pytest.approx(s2, rel=1e-12) and np.count_nonzero(np.tril(JJ, -1)) == 0


@pytest.mark.skipif(not RefFopt.available(), reason="oracle/_ref/libref_fopt.so needs /root/reference at build time")
def test_nonrigid_normals_and_buckets_against_reference_header():
    """Non-rigid mode (OptApp.cpp:120-206): UpdateAllNormal bit for bit, the two 24-entry buckets exactly (they are plain
    products, no Eigen reductions), and the merged triplets against a numpy construction from the buckets."""
    sc = make_scene(num=3, n=5000, res=4)
    own, ref = _load(sc, FoptOracle), _load(sc, RefFopt)
    ctr = lattice_ctr(sc["num"], sc["res"], sc["length"], [np.eye(4)] * sc["num"], 0.004, np.random.default_rng(2))
    for f in range(sc["num"]):
        own.update_normals(f, ctr[f * own.nper:(f + 1) * own.nper])
        ref.update_normals(f, ctr)
        a, b = own.points(f), ref.points(f)
        assert np.array_equal(a["n"].view(np.uint32), b["n"].view(np.uint32)) and np.array_equal(a["p"], b["p"])
    rng = np.random.default_rng(8)
    for i, j, pr in sc["pairs"]:
        for k in rng.choice(pr.shape[0], 100, replace=False):
            A = own.nonrigid_bucket(i, int(pr[k, 0]), j, int(pr[k, 1]), 1.7)
            B = ref.nonrigid_bucket(i, int(pr[k, 0]), j, int(pr[k, 1]), 1.7)
            for x, y in zip(A, B):
                assert np.array_equal(x, y)
    sub = [(i, j, pr[:150]) for i, j, pr in sc["pairs"]]
    own.set_pairs(sub)
    rows, cols, vals = own.assemble_nonrigid(1.7)
    M = own.nper * sc["num"]
    D = np.zeros((M, M))
    for i, j, pr in sub:
        for a, c in pr:
            i1, v1, i2, v2 = own.nonrigid_bucket(i, int(a), j, int(c), 1.7)
            r1, r2 = i * own.nper + i1, j * own.nper + i2
            np.add.at(D, (r1[:, None], r1[None, :]), np.outer(v1, v1))
            np.add.at(D, (r2[:, None], r2[None, :]), np.outer(v2, v2))
            np.add.at(D, (r1[:, None], r2[None, :]), np.outer(v1, v2))
    S = np.zeros((M, M))
    S[rows, cols] = vals
    assert np.allclose(S, D, rtol=1e-11, atol=1e-13) and rows.size == np.count_nonzero(D) + int((D[rows, cols] == 0).sum())
