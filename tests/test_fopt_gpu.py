"""SURVEY.md 8f-2 on the GPU: er_fopt_* (elasticreconstruction_amd/csrc/er_fopt.hip) against oracle/fopt_oracle.cpp.
Bars: float32 point state (GetCoordinate, UpdatePose, UpdateAllPointPN) BIT-EXACT; assembled float64 matrices within
1e-11 relative of the sequential CPU sums (FP64 matrix-core accumulation + atomics in another order); the rigid
optimisation closed with a dense solve must pull noisy poses back onto the ground truth."""
import numpy as np
import pytest

from elasticreconstruction_amd.fopt import FragmentOptimizer
from oracle.pyoracle import FoptOracle
from fopt_helpers import lattice_ctr, make_scene

pytestmark = pytest.mark.gpu


def _both(sc):
    g, o = FragmentOptimizer(sc["num"], sc["res"], sc["length"]), FoptOracle(sc["num"], sc["res"], sc["length"])
    for f, (x, n) in enumerate(sc["frags"]):
        assert g.SetCloud(f, x, n) == -1 and o.set_cloud(f, x, n) == -1
    return g, o


def _same_state(g, o, num):
    for f in range(num):
        a, b = g.points(f), o.points(f)
        for key in ("idx0", "val", "nval", "p", "n"):
            assert np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)), (f, key)


def _close(A, B, what):
    scale = np.abs(B).max()
    err = np.abs(A - B).max()
    assert err <= 1e-11 * scale, "%s: max abs error %.3g against scale %.3g" % (what, err, scale)


def test_point_state_bit_exact(gpu):
    sc = make_scene(num=3, n=30000)
    g, o = _both(sc)
    _same_state(g, o, sc["num"])
    for f in range(sc["num"]):
        M = sc["init"][f].astype(np.float32)
        g.UpdatePose(f, M)
        o.update_pose(f, M)
    _same_state(g, o, sc["num"])
    ctr = lattice_ctr(sc["num"], sc["res"], sc["length"], sc["init"], 0.003, np.random.default_rng(9))
    g.UpdateAllPointPN(ctr)
    for f in range(sc["num"]):
        o.update_point_pn(f, ctr[f * o.nper:(f + 1) * o.nper])
    _same_state(g, o, sc["num"])
    x, n = sc["frags"][0]
    xb = x.copy()
    xb[17, 2] = 3.5
    assert g.SetCloud(0, xb, n) == 17 and g.points(0)["p"].shape[0] == 17      # loading stops at the first out-of-bound point


def test_rigid_and_slac_assembly_match_oracle(gpu):
    sc = make_scene(num=4, n=40000)
    g, o = _both(sc)
    for f in range(sc["num"]):
        M = sc["init"][f].astype(np.float32)
        g.UpdatePose(f, M)
        o.update_pose(f, M)
    ngroups = g.SetCorrespondences(sc["pairs"])
    o.set_pairs(sc["pairs"])
    assert ngroups > len(sc["pairs"]) and sum(p[2].shape[0] for p in sc["pairs"]) > 20000
    JJ, Jb, s = g.AssembleRigid()
    J0, b0, s0 = o.assemble_rigid()
    _close(JJ, J0, "rigid JJ"); _close(Jb, b0, "rigid Jb")
    assert s == pytest.approx(s0, rel=1e-11) and np.allclose(JJ, JJ.T, rtol=1e-12, atol=1e-12 * np.abs(JJ).max())
    Rt = np.stack([P[:3, :3].T.reshape(9) for P in sc["init"]])
    JJ, Jb, s = g.AssembleSLAC(Rt)
    J0, b0, s0 = o.assemble_slac(Rt)
    _close(JJ, J0, "SLAC JJ"); _close(Jb, b0, "SLAC Jb")
    assert s == pytest.approx(s0, rel=1e-11) and np.count_nonzero(np.tril(JJ, -1)) == 0
    assert np.array_equal(JJ != 0, J0 != 0)                                  # same sparsity pattern
    # after a lattice update (SLAC's per-iteration state change) and with an empty pair in the list
    ctr = lattice_ctr(sc["num"], sc["res"], sc["length"], sc["init"], 0.002, np.random.default_rng(4))
    g.UpdateAllPointPN(ctr)
    for f in range(sc["num"]):
        o.update_point_pn(f, ctr[f * o.nper:(f + 1) * o.nper])
    pairs = sc["pairs"][:3] + [(0, 3, np.zeros((0, 2), np.int32))]
    g.SetCorrespondences(pairs)
    o.set_pairs(pairs)
    JJ, Jb, s = g.AssembleSLAC(Rt)
    J0, b0, s0 = o.assemble_slac(Rt)
    _close(JJ, J0, "SLAC JJ (2)"); _close(Jb, b0, "SLAC Jb (2)")
    with pytest.raises(Exception):
        g.SetCorrespondences([(0, 1, np.array([[10 ** 8, 0]], np.int32))])
    # the factor of er_fopt_factor_slac lives in the matrix the assembly calls fill: a later assembly must invalidate it
    g.FactorSLAC(Rt, 1000.0)
    x = g.Solve(np.ones(6 * sc["num"] + g.nper_))
    assert np.all(np.isfinite(x))
    g.AssembleSLAC(Rt)
    with pytest.raises(Exception, match="no factored system"):
        g.Solve(np.ones(6 * sc["num"] + g.nper_))
    g.FactorSLAC(Rt, 1000.0)
    assert np.allclose(g.Solve(np.ones(6 * sc["num"] + g.nper_)), x, rtol=1e-9, atol=1e-12)
    g.AssembleRigid()
    with pytest.raises(Exception, match="no factored system"):
        g.Solve(np.ones(6 * sc["num"] + g.nper_))


def test_rigid_optimisation_recovers_poses(gpu):
    sc = make_scene(num=3, n=40000)
    g, _ = _both(sc)
    g.SetCorrespondences(sc["pairs"])
    pose, scores = g.OptimizeRigid(sc["init"], max_iteration=4)
    assert scores[-1] < 0.2 * scores[0]
    for f in range(1, sc["num"]):                                            # fragment 0 is the gauge (poses[0] == init[0])
        rel_before = np.linalg.inv(sc["poses"][0]) @ sc["poses"][f]
        rel_init = np.linalg.inv(sc["init"][0]) @ sc["init"][f]
        rel_after = np.linalg.inv(pose[0]) @ pose[f]
        assert np.abs(rel_after - rel_before).max() < 0.25 * np.abs(rel_init - rel_before).max()


def test_nonrigid_assembly_matches_oracle(gpu):
    """Non-rigid mode (OptApp.cpp:120-206): UpdateAllNormal bit-exact; the block-sparse data term merged to triplets
    equals the oracle's merged triplets (same nonzero set, values within 1e-11 of the largest entry)."""
    sc = make_scene(num=3, n=30000)
    g, o = _both(sc)
    ctr = lattice_ctr(sc["num"], sc["res"], sc["length"], [np.eye(4)] * sc["num"], 0.004, np.random.default_rng(2))
    g.UpdateAllNormal(ctr)
    for f in range(sc["num"]):
        o.update_normals(f, ctr[f * o.nper:(f + 1) * o.nper])
    _same_state(g, o, sc["num"])
    pairs = [(i, j, pr[:6000]) for i, j, pr in sc["pairs"]]
    ng = g.SetCorrespondences(pairs)
    o.set_pairs(pairs)
    r, c, v = g.NonrigidTriplets(1.0)
    r0, c0, v0 = o.assemble_nonrigid(1.0)
    M = o.nper * sc["num"]
    k, k0 = r * M + c, r0 * M + c0
    keep = v != 0                                              # blocks are dense 24x24; the oracle only holds touched entries
    assert set(k[keep].tolist()) <= set(k0.tolist())
    d = dict(zip(k.tolist(), v.tolist()))
    got = np.array([d.get(int(q), 0.0) for q in k0])
    assert np.abs(got - v0).max() <= 1e-11 * np.abs(v0).max()
    assert ng > 50
    diag, off, info = g.AssembleNonrigid(1.0)
    assert off.shape == (ng, 24, 24) and info.shape == (ng, 4) and np.allclose(diag, diag.transpose(0, 1, 3, 2), rtol=1e-12, atol=1e-12 * np.abs(diag).max())


# ---- end to end against the reference PROGRAM ---------------------------------------------------------------------
import os

from test_fopt_oracle import REF_BIN, _run_ref, _write_dataset


def _read_ctr(fn):
    return np.loadtxt(fn).reshape(-1)


def _read_log(fn):
    rows = [l.split() for l in open(fn) if l.strip()]
    return [np.array(rows[k + 1:k + 5], np.float64) for k in range(0, len(rows), 5)]


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/FragmentOptimizer_ref is built where /root/reference exists")
def test_optimisation_loops_match_the_reference_program(gpu, tmp_path):
    """fopt.FragmentOptimizer (GPU assembly + host regularizer / dense solve) against the reference's own FragmentOptimizer
    program (oracle/_ref, CHOLMOD replaced by the dense shim) on the same files: output.ctr and pose.log after two outer
    iterations of --rigid and --slac and one of the default non-rigid mode.  Tolerances are those of two different
    Cholesky solvers iterated twice (1e-7 absolute on metre-scale coordinates; the files carry 8-10 decimals)."""
    d = str(tmp_path)
    sc = make_scene(num=3, n=3000)
    poses = _write_dataset(sc, d)
    num = sc["num"]

    def fresh(s):
        g = FragmentOptimizer(s["num"], s["res"], s["length"])
        for f, (x, n) in enumerate(s["frags"]):
            assert g.SetCloud(f, x, n) == -1
        g.SetCorrespondences(s["pairs"])
        return g

    args = ["--num", str(num), "--resolution", "8", "--length", "3.0", "--iteration", "2"]
    # rigid
    _run_ref(d, "rigid", "reg_output.log", args)
    ref_pose, ref_ctr = _read_log(os.path.join(d, "pose.log")), _read_ctr(os.path.join(d, "out_rigid.ctr"))
    g = fresh(sc)
    pose, _ = g.OptimizeRigid(poses, max_iteration=2)
    assert max(np.abs(a - b).max() for a, b in zip(pose, ref_pose)) < 1e-7
    lat = g._canonical_lattice().reshape(-1, 3)
    ctr = np.concatenate([g._apply(P, lat).reshape(-1) for P in pose])           # Pose2Ctr, OptApp.cpp:735-750
    assert np.abs(ctr - ref_ctr).max() < 1e-7
    # SLAC
    _run_ref(d, "slac", "reg_output.log", args)
    ref_pose, ref_ctr = _read_log(os.path.join(d, "pose.log")), _read_ctr(os.path.join(d, "out_slac.ctr"))
    for solver in ("device", "host"):                                        # dense Cholesky in HBM (rocSOLVER) / numpy on the host
        g = fresh(sc)
        pose, expand, _ = g.OptimizeSLAC(poses, weight=1.0, max_iteration=2, solver=solver)
        assert max(np.abs(a - b).max() for a, b in zip(pose, ref_pose)) < 1e-7, solver
        assert np.abs(expand - ref_ctr).max() < 1e-7, solver
    # non-rigid (default mode), resolution 4 keeps the dense solves quick
    sc4 = make_scene(num=3, n=3000, res=4)
    d4 = os.path.join(d, "r4")
    os.makedirs(d4)
    poses4 = _write_dataset(sc4, d4)
    _run_ref(d4, "nonrigid", "reg_output.log", ["--num", str(num), "--resolution", "4", "--length", "3.0", "--weight", "1.7", "--inner_iteration", "2"])
    ref_ctr = _read_ctr(os.path.join(d4, "out_nonrigid.ctr"))
    for solver in ("device", "host"):
        g = fresh(sc4)
        ctr, _ = g.OptimizeNonrigid(poses4, weight=1.7, max_iteration=1, max_inner_iteration=2, solver=solver)
        assert np.abs(ctr - ref_ctr).max() < 1e-6, solver


def test_block_sparse_cholesky_equals_dense(gpu, monkeypatch):
    """The non-rigid system factored as the block-sparse lower triangle of fragment blocks (ER_FOPT_DENSE_MAX=0) against the
    dense factorisation of the same system: 5 fragments whose pair graph is a ring with one chord -- (0,1) (1,2) (2,3) (3,4)
    (0,4) (1,3) -- so the symbolic pass has to create fill-in blocks ((4,1), (4,2), (4,3) ...) that no correspondence
    list touches.  The solutions of three right-hand sides must agree to 1e-9 relative."""
    sc = make_scene(num=5, n=12000)
    keep = {(0, 1), (1, 2), (2, 3), (3, 4), (0, 4), (1, 3)}
    pairs = [(i, j, pr[:3000]) for i, j, pr in sc["pairs"] if (i, j) in keep]
    assert len(pairs) == len(keep) and all(len(pr) > 500 for _, _, pr in pairs)
    rng = np.random.default_rng(4)
    ctr = lattice_ctr(sc["num"], sc["res"], sc["length"], [np.eye(4)] * sc["num"], 0.003, rng)
    M = 2187 * sc["num"]
    rhs = [rng.normal(size=M) for _ in range(3)]
    sols = {}
    for mode, dense_max in (("dense", "1000000"), ("blocked", "0")):
        monkeypatch.setenv("ER_FOPT_DENSE_MAX", dense_max)
        g = FragmentOptimizer(sc["num"], sc["res"], sc["length"])
        for f, (x, n) in enumerate(sc["frags"]):
            assert g.SetCloud(f, x, n) == -1
        g.UpdateAllNormal(ctr)
        g.SetCorrespondences(pairs)
        g.FactorNonrigid(1.0)
        sols[mode] = [g.Solve(b) for b in rhs]
        g.FactorNonrigid(2.5)                                   # a second factorisation in the same handle (buffers reused)
        sols[mode].append(g.Solve(rhs[0]))
        g.close()
    for a, b in zip(sols["dense"], sols["blocked"]):
        assert np.isfinite(b).all()
        assert np.abs(a - b).max() <= 1e-9 * np.abs(a).max(), "block-sparse and dense solutions differ by %.3g (scale %.3g)" % (np.abs(a - b).max(), np.abs(a).max())
    assert np.abs(sols["dense"][0] - sols["dense"][3]).max() > 1e-6 * np.abs(sols["dense"][0]).max()      # the weight does matter


def _pivot(err):
    """1-based index of the first non-positive pivot as the library reports it (+ the host Cholesky's, under ER_FOPT_DIAG)."""
    import re
    m = re.search(r"not positive definite \(Cholesky pivot (\d+)", str(err))
    assert m, str(err)
    h = re.search(r"host Cholesky of the SAME assembled matrix: (.*?) \(pivot (\d+)\)", str(err))
    return int(m.group(1)), (int(h.group(2)), h.group(1)) if h else None


def test_cholesky_failure_path_reports_the_first_non_positive_pivot(gpu, monkeypatch):
    """The failure path of potrf_lower (the library's own recursive blocked Cholesky) -- what the reference gets from CHOLMOD's status
    (FragmentOptimizer/OptApp.cpp:209-211, 389-393).  A positive definite system A becomes indefinite at a KNOWN place when a large
    negative number is added to one diagonal entry g (er_fopt_debug_shift_diagonal): the leading g x g block is untouched, so pivots
    1 .. g succeed and pivot g + 1 is the first non-positive one -- in the first 64 x 64 diagonal block, deep in the matrix (the j_base
    arithmetic of potrf_rec across recursion levels), in the last fragment block (the block-sparse layout's global index), and for a NaN
    entry.  Dense layout: the reported index must also equal the one a plain host Cholesky of the SAME assembled matrix finds
    (ER_FOPT_DIAG).  Afterwards the handle refactors cleanly and reproduces its first solution.  (Round 3's version of this test used a
    negative DATA weight: that weight multiplies the Jacobian rows, the Hessian sees its square, and the system stays positive
    definite -- checked here as such with a weight of -3; at -1e6 the data term outweighs the regulariser by 1e12 and the
    factorisation fails for a different reason, float64 cancellation, in the block-sparse layout: pivot 6549 of 6561.)"""
    from elasticreconstruction_amd._ffi import ErError
    sc = make_scene(num=3, n=6000)
    rng = np.random.default_rng(11)
    ctr = lattice_ctr(sc["num"], sc["res"], sc["length"], [np.eye(4)] * sc["num"], 0.003, rng)
    M = 2187 * sc["num"]
    b = rng.normal(size=M)
    for layout, dense_max in (("dense", "1000000"), ("blocked", "0")):
        monkeypatch.setenv("ER_FOPT_DENSE_MAX", dense_max)
        monkeypatch.delenv("ER_FOPT_DIAG", raising=False)
        g = FragmentOptimizer(sc["num"], sc["res"], sc["length"])
        for f, (x, n) in enumerate(sc["frags"]):
            assert g.SetCloud(f, x, n) == -1
        g.UpdateAllNormal(ctr)
        g.SetCorrespondences(sc["pairs"])
        g.FactorNonrigid(1.0)
        x1, x2 = g.Solve(b), g.Solve(2.0 * b)
        assert np.isfinite(x1).all() and np.abs(x2 - 2.0 * x1).max() <= 1e-12 * np.abs(x1).max()
        g.FactorNonrigid(3.0)
        x3 = g.Solve(b)
        g.FactorNonrigid(-3.0)                                   # the weight multiplies the Jacobian ROWS: the Hessian sees its square, the
        xm = g.Solve(b)                                          # system is the one of +3 -- positive definite, same solution, no error
        assert np.isfinite(xm).all() and np.abs(xm - x3).max() <= 1e-9 * np.abs(x3).max() and np.abs(x3 - x1).max() > 1e-6 * np.abs(x1).max()
        for idx, value in ((5, -1.0e12), (63, -1.0e12), (64, -1.0e12), (1000, -1.0e12), (2187 + 1093, -1.0e12), (2 * 2187 + 700, -1.0e12),
                           (M - 1, -1.0e12), (4500, float("nan"))):
            g.DebugShiftDiagonal(idx, value)
            with pytest.raises(ErError, match="not positive definite") as e:
                g.FactorNonrigid(1.0)
            assert _pivot(e.value)[0] == idx + 1, (layout, idx, str(e.value))
            with pytest.raises(ErError, match="no factored system"):
                g.Solve(b)                                       # a failed factorisation leaves nothing to solve with
        if layout == "dense":                                    # ... and the host Cholesky of the same matrix agrees on the index
            monkeypatch.setenv("ER_FOPT_DIAG", "1")
            for idx in (70, 3000):
                g.DebugShiftDiagonal(idx, -1.0e12)
                with pytest.raises(ErError, match="not positive definite") as e:
                    g.FactorNonrigid(1.0)
                dev, host = _pivot(e.value)
                assert dev == idx + 1 and host is not None and host[0] == dev and host[1].startswith("ALSO not positive definite"), str(e.value)
            monkeypatch.delenv("ER_FOPT_DIAG")
        g.DebugShiftDiagonal(-1, 0.0)
        g.FactorNonrigid(1.0)                                    # the handle is still usable afterwards
        assert np.abs(g.Solve(b) - x1).max() <= 1e-12 * np.abs(x1).max()
        g.close()


def test_slac_with_a_negative_regulariser_is_reported_not_positive_definite(gpu, monkeypatch):
    """er_fopt_factor_slac scales the lattice Laplacian and the anchor by default_weight (OptApp.cpp:452-464): a large negative value
    makes thisJJ indefinite in its lattice part.  The library must say so, name the same first pivot as a host Cholesky of the same
    matrix, and factor the well-posed system afterwards."""
    from elasticreconstruction_amd._ffi import ErError
    sc = make_scene(num=3, n=6000)
    g = FragmentOptimizer(sc["num"], sc["res"], sc["length"])
    for f, (x, n) in enumerate(sc["frags"]):
        assert g.SetCloud(f, x, n) == -1
        g.UpdatePose(f, sc["init"][f].astype(np.float32))
    g.SetCorrespondences(sc["pairs"])
    Rt = np.stack([P[:3, :3].T.reshape(9) for P in sc["init"]])
    N = 6 * sc["num"] + g.nper_
    Jb, _ = g.FactorSLAC(Rt, 1000.0)
    x1 = g.Solve(Jb)
    monkeypatch.setenv("ER_FOPT_DIAG", "1")
    with pytest.raises(ErError, match="not positive definite") as e:
        g.FactorSLAC(Rt, -1.0e6)
    dev, host = _pivot(e.value)
    assert 6 * sc["num"] < dev <= N and host is not None and host[0] == dev, str(e.value)     # the pose block (gauge + data) is fine
    monkeypatch.delenv("ER_FOPT_DIAG")
    with pytest.raises(ErError, match="not positive definite") as e:
        g.FactorSLAC(Rt, -1.0e6)
    assert _pivot(e.value) == (dev, None)
    Jb2, _ = g.FactorSLAC(Rt, 1000.0)
    assert np.array_equal(Jb, Jb2) or np.allclose(Jb, Jb2, rtol=1e-12, atol=1e-12 * np.abs(Jb).max())
    assert np.abs(g.Solve(Jb2) - x1).max() <= 1e-9 * np.abs(x1).max()
    g.close()


def test_device_resident_correspondence_hand_off(gpu):
    """VERDICT round 4 (5, 7): BuildCorrespondence's lists reach FragmentOptimizer WITHOUT crossing PCIe twice (CorresApp.cpp:175-184 writes what
    OptApp.cpp:100-118 reads).  er_registration_batch is handed list buffers that live in HBM (er_device_alloc) -- the lists are then copied device
    to device -- and er_fopt_set_correspondences_dev sorts them by lattice cell pair on the GPU.  Against the host path on the same pairs:
    transforms, counts and information matrices identical, every downloaded list identical to the host list, the same (pair, cell, cell) groups
    in the same order, assembled rigid and SLAC systems equal to 1e-12 (the assembly adds with float64 atomics: no run-to-run bit equality
    even on one path), and an out-of-range row is refused on the device."""
    import ctypes as C
    from elasticreconstruction_amd import _ffi, synth
    from elasticreconstruction_amd.icp import Cloud, DeviceLists, registration_batch, registration_batch_dev
    frs = synth.fragment_set(4, 60000, device="cuda:0")
    clouds = [Cloud(x, n, 0.03) for x, n, _ in frs]
    ids = [(0, 1), (1, 2), (2, 3), (0, 2), (1, 3), (0, 3)]
    Ts = [np.linalg.inv(frs[a][2]) @ frs[b][2] @ synth.perturbation(40 + k, 2.0, 0.02) for k, (a, b) in enumerate(ids)]
    srcs, tgts = [clouds[b] for _, b in ids], [clouds[a] for a, _ in ids]
    host = registration_batch(srcs, tgts, Ts, want_info=True)
    dl = DeviceLists(srcs)
    dev = registration_batch_dev(srcs, tgts, Ts, dl, want_info=True)
    assert np.array_equal(dev["counts"], host["counts"]) and np.array_equal(dev["accepted"], host["accepted"]) and host["accepted"].all()
    assert np.array_equal(dev["iterations"], host["iterations"]) and np.array_equal(dev["T"].view(np.uint32), host["T"].view(np.uint32))
    assert np.allclose(dev["info"], host["info"], rtol=1e-12, atol=0)
    assert [int(c) for c in dl.counts] == [len(l) for l in host["lists"]] and min(dl.counts) > 10000
    for k in range(len(ids)):
        assert np.array_equal(dl.download(k), host["lists"][k]), "list %d differs between the host and the device-resident path" % k
    g1, g2 = FragmentOptimizer(4, 8, 3.0), FragmentOptimizer(4, 8, 3.0)
    for g in (g1, g2):
        for f, (x, n, _) in enumerate(frs):
            assert g.SetCloud(f, x, n) == -1
    n1 = g1.SetCorrespondences([(a, b, host["lists"][k]) for k, (a, b) in enumerate(ids)])
    n2 = g2.SetCorrespondencesDev(ids, dl)
    assert n1 == n2 > len(ids)
    gi1, gi2 = np.zeros(4 * n1, np.int32), np.zeros(4 * n2, np.int32)
    g1._lib.er_fopt_group_info(g1._h, _ffi.ptr(gi1))
    g2._lib.er_fopt_group_info(g2._h, _ffi.ptr(gi2))
    assert np.array_equal(gi1, gi2), "the (pair, cell, cell) groups differ"
    (J1, b1, s1), (J2, b2, s2) = g1.AssembleRigid(), g2.AssembleRigid()
    assert np.allclose(J1, J2, rtol=0, atol=1e-12 * np.abs(J1).max()) and np.allclose(b1, b2, rtol=0, atol=1e-12 * np.abs(b1).max()) and s1 == pytest.approx(s2, rel=1e-12)
    Rt = np.stack([np.eye(3).reshape(9) for _ in range(4)])
    (J1, b1, s1), (J2, b2, s2) = g1.AssembleSLAC(Rt), g2.AssembleSLAC(Rt)
    assert np.allclose(J1, J2, rtol=0, atol=1e-12 * np.abs(J1).max()) and np.allclose(b1, b2, rtol=0, atol=1e-12 * np.abs(b1).max())
    assert np.array_equal(J1 != 0, J2 != 0)
    # an empty list in the middle, then a row that points outside its fragment: refused by the range check ON THE DEVICE
    dl.counts[1] = 0
    assert g2.SetCorrespondencesDev(ids, dl) < n2
    bad = np.array([[len(frs[0][0]) + 5, 0]], np.int32)
    _ffi.check(g2._lib.er_host_copy_h2d(C.c_void_p(dl.base + 4 * int(dl.offs[1])), _ffi.ptr(bad), 8), "er_host_copy_h2d")
    dl.counts[1] = 1
    with pytest.raises(Exception, match="out of range"):
        g2.SetCorrespondencesDev(ids, dl)
    dl.close()
