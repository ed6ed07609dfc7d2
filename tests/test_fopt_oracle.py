"""SURVEY.md 8f-2, CPU side: oracle/fopt_oracle.cpp (plain C++ restatement of FragmentOptimizer's point updates and
Hessian assembly) pinned against the reference: float32 point state bit for bit against oracle/_ref/libref_fopt.so (the
reference's own PointCloud.{h,cpp} compiled in place), the assembled systems of all three modes against what the
reference PROGRAM (oracle/_ref/FragmentOptimizer_ref, compiled in place on a CHOLMOD shim) hands to its solver, and the
dense assembly against an independent numpy construction from the oracle's buckets."""
import numpy as np
import pytest

from oracle.pyoracle import FoptOracle, RefFopt
from elasticreconstruction_amd import synth
from fopt_helpers import lattice_ctr, make_scene


def _load(sc, cls):
    o = cls(sc["num"], sc["res"], sc["length"])
    for f, (x, n) in enumerate(sc["frags"]):
        assert o.set_cloud(f, x, n) == -1
    return o


@pytest.mark.skipif(not RefFopt.available(), reason="oracle/_ref/libref_fopt.so needs /root/reference at build time")
def test_point_state_bitwise_against_reference_header():
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
def test_nonrigid_normals_against_reference_header_and_triplets():
    """Non-rigid mode (OptApp.cpp:120-206): UpdateAllNormal bit for bit against the reference's PointCloud, and the merged
    triplets against a numpy construction from the oracle's own buckets (the loop itself is pinned to the reference program
    in test_assembly_pinned_to_the_reference_program)."""
    sc = make_scene(num=3, n=5000, res=4)
    own, ref = _load(sc, FoptOracle), _load(sc, RefFopt)
    ctr = lattice_ctr(sc["num"], sc["res"], sc["length"], [np.eye(4)] * sc["num"], 0.004, np.random.default_rng(2))
    for f in range(sc["num"]):
        own.update_normals(f, ctr[f * own.nper:(f + 1) * own.nper])
        ref.update_normals(f, ctr)
        a, b = own.points(f), ref.points(f)
        assert np.array_equal(a["n"].view(np.uint32), b["n"].view(np.uint32)) and np.array_equal(a["p"], b["p"])
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


# ---- end-to-end pin against the reference PROGRAM ---------------------------------------------------------------
import os
import subprocess

REF_BIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "FragmentOptimizer_ref")


def _write_dataset(sc, d):
    """The files the reference program reads: cloud_bin_xyzn_<i>.xyzn, corres_<i>_<j>.txt, reg_output.log, ipose.log.
    Returns the poses as the program will parse them (8 decimals)."""
    for f, (x, n) in enumerate(sc["frags"]):
        with open(os.path.join(d, "cloud_bin_xyzn_%d.xyzn" % f), "w") as fh:
            for p, q in zip(x, n):
                fh.write("%.9g %.9g %.9g %.9g %.9g %.9g\n" % (p[0], p[1], p[2], q[0], q[1], q[2]))     # 9 digits round-trip float32
    with open(os.path.join(d, "reg_output.log"), "w") as fh:
        for i, j, pr in sc["pairs"]:
            np.savetxt(os.path.join(d, "corres_%d_%d.txt" % (i, j)), pr, fmt="%d")
            fh.write("%d\t%d\t%d\n" % (i, j, pr.shape[0]))
            for r in np.eye(4):
                fh.write("%.8f %.8f %.8f %.8f\n" % tuple(r))
    with open(os.path.join(d, "reg_empty.log"), "w") as fh:      # same pairs, all invalid (frame_ == -1): no data term
        for i, j, pr in sc["pairs"]:
            fh.write("%d\t%d\t-1\n" % (i, j))
            for r in np.eye(4):
                fh.write("%.8f %.8f %.8f %.8f\n" % tuple(r))
    # initial poses through --rgbdslam (InitIPose, OptApp.cpp:49-72: ipose = basepose * traj[0]^-1 * traj[i * interval] * basepose^-1);
    # --ipose cannot be used: with it InitIPose returns before pose_ / pose_rot_t_ are sized and the program writes out of bounds.
    base = synth.basepose(sc["length"])
    G = []
    with open(os.path.join(d, "rgbd.log"), "w") as fh:
        for f, P in enumerate(sc["init"]):
            g = np.linalg.inv(base) @ P @ base
            fh.write("%d\t%d\t%d\n" % (f, f, f + 1))
            rows = ["%.8f %.8f %.8f %.8f" % tuple(r) for r in g]
            fh.write("\n".join(rows) + "\n")
            G.append(np.array([[float(v) for v in r.split()] for r in rows]))
    left = base @ np.linalg.inv(G[0])
    return [left @ g @ np.linalg.inv(base) for g in G]


def _run_ref(d, mode, reg, extra=()):
    env = dict(os.environ, ER_CHOLMOD_DUMP=os.path.join(d, "dump_" + mode + "_" + reg.split(".")[0]), ER_ORACLE_QUIET="1", OMP_NUM_THREADS="8",
               OMP_WAIT_POLICY="passive")
    cmd = [REF_BIN] + list(extra) + ["--registration", os.path.join(d, reg), "--dir", d + "/", "--rgbdslam", os.path.join(d, "rgbd.log"),
                                     "--interval", "1", "--blacklistpair", "0", "--iteration", "1", "--inner_iteration", "1",
                                     "--save_to", os.path.join(d, "out_%s.ctr" % mode)]      # the first occurrence of a flag wins
    if mode != "nonrigid":
        cmd.append("--" + mode)
    subprocess.run(cmd, check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env, timeout=180)
    return env["ER_CHOLMOD_DUMP"]


def _load_A(prefix, k=0):
    raw = open("%s_A%d.bin" % (prefix, k), "rb").read()
    n, nnz = np.frombuffer(raw[:16], np.int64)
    rec = np.frombuffer(raw[16:], dtype=np.dtype([("r", "<i4"), ("c", "<i4"), ("v", "<f8")]), count=int(nnz))
    A = np.zeros((int(n), int(n)))
    np.add.at(A, (rec["r"], rec["c"]), rec["v"])
    return A


def _load_b(prefix, k=0):
    raw = open("%s_b%d.bin" % (prefix, k), "rb").read()
    n = int(np.frombuffer(raw[:8], np.int64)[0])
    return np.frombuffer(raw[8:], np.float64, count=n).copy()


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/FragmentOptimizer_ref needs /root/reference at build time")
def test_assembly_pinned_to_the_reference_program(tmp_path):
    """The reference's own FragmentOptimizer (compiled in place, unmodified; CHOLMOD replaced by oracle/cholmod_shim.cpp)
    runs one iteration of each mode on files written here; the shim dumps the system it is asked to factorize / solve
    (thisJJ, thisJb, thisAA).  The restated assembly of oracle/fopt_oracle.cpp must reproduce them: rigid directly, SLAC and
    non-rigid after subtracting the regularizer-only system of a run whose pairs are all marked invalid."""
    d = str(tmp_path)
    sc = make_scene(num=3, n=2500)
    poses = _write_dataset(sc, d)
    num = sc["num"]
    args = ["--num", str(num), "--resolution", "8", "--length", "3.0"]

    # ---- rigid: thisJJ (upper triangle as the solver sees it) and thisJb, OptApp.cpp:312-393 -------------
    pre = _run_ref(d, "rigid", "reg_output.log", args)
    A, b = _load_A(pre), _load_b(pre)
    o = _load(sc, FoptOracle)
    for f in range(num):
        o.update_pose(f, poses[f].astype(np.float32))            # pose_[ i ].cast< float >(), :296
    o.set_pairs(sc["pairs"])
    JJ, Jb, _ = o.assemble_rigid()
    assert np.abs(np.triu(A) - np.triu(JJ)).max() <= 1e-12 * np.abs(JJ).max()
    assert np.abs(b - Jb).max() <= 1e-12 * np.abs(Jb).max()

    # ---- SLAC: data term = (full system) - (regularizer-only system), OptApp.cpp:449-560 ------------------
    pre = _run_ref(d, "slac", "reg_output.log", args)
    pre0 = _run_ref(d, "slac", "reg_empty.log", args)
    A, A0, b = _load_A(pre), _load_A(pre0), _load_b(pre)
    Rt = np.stack([P[:3, :3].T.reshape(9) for P in poses])     # pose_rot_t_[ i ] = pose_[ i ].block<3,3>(0,0).transpose(), :443
    JJ, Jb, _ = o.assemble_slac(Rt)
    D = np.triu(A) - np.triu(A0)
    assert np.abs(D - JJ).max() <= 1e-10 * np.abs(JJ).max()
    assert np.abs(b - Jb).max() <= 1e-10 * np.abs(Jb).max()       # baseJb vanishes at the initial lattice (tempCtr == ictr)

    # ---- non-rigid (resolution 4 keeps the dense shim quick): thisAA - baseAA, OptApp.cpp:120-211 --------
    sc4 = make_scene(num=3, n=2500, res=4)
    d4 = os.path.join(d, "r4")
    os.makedirs(d4)
    poses4 = _write_dataset(sc4, d4)
    args4 = ["--num", str(num), "--resolution", "4", "--length", "3.0", "--weight", "1.7"]
    pre = _run_ref(d4, "nonrigid", "reg_output.log", args4)
    pre0 = _run_ref(d4, "nonrigid", "reg_empty.log", args4)
    A, A0 = _load_A(pre), _load_A(pre0)
    o4 = _load(sc4, FoptOracle)
    # InitCtr, OptApp.cpp:709-721: ctr = ipose_[ l ] * ( i, j, k ) * unit_length_  (Matrix4d * Vector4d, column by column)
    ul = 3.0 / 4
    ctr = []
    for P in poses4:
        for k in range(5):
            for j in range(5):
                for i in range(5):
                    pos = (i * ul, j * ul, k * ul)
                    ctr.extend([((P[r, 0] * pos[0] + P[r, 1] * pos[1]) + P[r, 2] * pos[2]) + P[r, 3] * 1.0 for r in range(3)])
    ctr = np.array(ctr)
    for f in range(num):
        o4.update_normals(f, ctr[f * o4.nper:(f + 1) * o4.nper])  # UpdateAllNormal( ctr ), :151-153
    o4.set_pairs(sc4["pairs"])
    r, c, v = o4.assemble_nonrigid(1.7)
    M = o4.nper * num
    S = np.zeros((M, M))
    S[r, c] = v
    D = np.triu(A) - np.triu(A0)
    assert np.abs(D - np.triu(S)).max() <= 1e-9 * np.abs(S).max()


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/FragmentOptimizer_ref needs /root/reference at build time")
def test_host_side_lattice_pieces_against_the_reference_program(tmp_path):
    """fopt.Lattice (pure numpy, what the Python mirror and -- restated in C++ -- bin/FragmentOptimizer keep on the host) against
    the systems the reference program assembles when every pair is invalid: baseJJ * default_weight + gauge (SLAC, InitBaseJJ
    OptApp.cpp:811-846) and baseAA (non-rigid, InitBaseAA :765-810); and the regularizer right-hand side of the non-rigid mode's
    first inner iteration (GetRotation :850-871 on an undeformed lattice gives the identity, so Ab = Laplacian * ctr)."""
    from elasticreconstruction_amd.fopt import Lattice
    d = str(tmp_path)
    sc = make_scene(num=3, n=800, res=4)
    poses = _write_dataset(sc, d)
    num = 3
    args = ["--num", "3", "--resolution", "4", "--length", "3.0", "--weight", "2.5"]
    L = Lattice(4, 3.0)
    lap = L.laplacian()
    # SLAC base system
    A0 = _load_A(_run_ref(d, "slac", "reg_empty.log", args))
    N = 6 * num + L.nper_
    B = np.zeros((N, N))
    B[6 * num:, 6 * num:] = lap
    anchor = 6 * num + L.GetIndex(2, 2, 0) * 3
    B[anchor:anchor + 3, anchor:anchor + 3] += np.eye(3)
    B *= num * 2.5                                             # default_weight = num_ * weight_, :421,:452
    B[np.arange(6), np.arange(6)] += 1.0                       # :459-464
    assert np.abs(np.triu(A0) - np.triu(B)).max() <= 1e-12 * np.abs(B).max()
    # non-rigid base system and first right-hand side
    pre = _run_ref(d, "nonrigid", "reg_empty.log", args)
    A0, b0 = _load_A(pre), _load_b(pre)
    M = num * L.nper_
    B = np.zeros((M, M))
    for l in range(num):
        B[l * L.nper_:(l + 1) * L.nper_, l * L.nper_:(l + 1) * L.nper_] = lap
    B[np.arange(3), np.arange(3)] += 1.0                       # :803-807
    assert np.abs(np.triu(A0) - np.triu(B)).max() <= 1e-12
    ctr = np.concatenate([Lattice.apply(P, L.canonical().reshape(-1, 3)).reshape(-1) for P in poses])
    Ab = np.zeros(M)
    for l in range(num):
        cur = ctr[l * L.nper_:(l + 1) * L.nper_].reshape(-1, 3)
        for v, nb, _ in L.edges():
            dif = cur[v] - cur[nb]
            R = Lattice.GetRotation(dif, dif)
            assert np.abs(R - np.eye(3)).max() < 1e-9
            bx = dif @ R.T
            Ab[l * L.nper_ + v * 3:l * L.nper_ + v * 3 + 3] += bx.sum(0)
            for t, w in enumerate(nb):
                Ab[l * L.nper_ + w * 3:l * L.nper_ + w * 3 + 3] -= bx[t]
    assert np.abs(Ab - b0).max() <= 1e-9 * max(np.abs(b0).max(), 1.0)
    # pose increments: AngleAxis(z) * AngleAxis(y) * AngleAxis(x) (:395-400)
    inc = Lattice.increment(np.array([0.1, -0.2, 0.3, 1.0, 2.0, 3.0]))
    assert np.allclose(inc[:3, :3] @ inc[:3, :3].T, np.eye(3), atol=1e-14) and np.allclose(inc[:3, 3], [1, 2, 3])
    assert inc[2, 0] == pytest.approx(-np.sin(-0.2)) and inc[1, 0] == pytest.approx(np.sin(0.3) * np.cos(-0.2))
