"""The drop-in boundary itself: the C++ host programs bin/Integrate and bin/BuildCorrespondence run on files
with the reference's own command lines and must produce the reference's files.
  * Integrate: world.pcd equals -- as a point set -- the output of the REFERENCE binary
    (oracle/_ref/Integrate_ref = /root/reference/Integrate/*.cpp compiled unmodified, fed the same raw depth
    stream through the stub grabber), bit for bit; falls back to the oracle port when that binary is absent.
  * BuildCorrespondence: against the REFERENCE program too (oracle/_ref/BuildCorrespondence_ref =
    /root/reference/BuildCorrespondence/*.cpp compiled in place against oracle/stub_corres; it travels to the GPU box) and
    against its committed outputs (tests/golden/corres_golden.json): -1 pattern and counts of the registration pass, transforms
    within 1e-5; then a FindCorrespondence-only pass from identical 8-decimal transforms where corres_<i>_<j>.txt must be
    byte-identical, frame fields equal and reg_output.info equal to the printed precision.  The Python mirror (same kernels)
    is compared byte for byte in a second test."""
import os
import subprocess

import numpy as np
import pytest

from elasticreconstruction_amd import formats, synth
from elasticreconstruction_amd.icp import CorresApp
from oracle import pyoracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "elasticreconstruction_amd", "bin")


def sorted_points(p):
    a = np.ascontiguousarray(p, np.float32).reshape(-1, 4)
    return a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]


def write_integrate_inputs(d, sc, depth):
    n, I = sc["n"], sc["interval"]
    num = n // I
    pose = [formats.FramedTransformation(i, i, i + 1, sc["pose"][i]) for i in range(num)]
    seg = [formats.FramedTransformation(i, i, i + 1, sc["seg"][i]) for i in range(n)]
    # F + 1 trajectory entries so that exactly F frames are integrated (SURVEY.md 8d frame-count convention)
    formats.save_log(os.path.join(d, "pose.log"), pose + [formats.FramedTransformation(num, num, num + 1, sc["pose"][-1])])
    formats.save_log(os.path.join(d, "seg.log"), seg + [formats.FramedTransformation(n + j, n + j, n + j + 1, sc["seg"][-1]) for j in range(I)])
    formats.save_ctr(os.path.join(d, "grids.ctr"), sc["grids"])
    depth.tofile(os.path.join(d, "frames.raw"))


def test_integrate_program_equals_reference_binary(gpu, tmp_path):
    d = str(tmp_path)
    sc = synth.make_scenario(12, interval=4, warp=True, amplitude=0.004, seed=21)
    depth = synth.to_numpy_u16(sc["depth"])
    write_integrate_inputs(d, sc, depth)
    args = ["--pose_traj", "pose.log", "--seg_traj", "seg.log", "--ctr", "grids.ctr", "--num", "3", "--resolution", "8",
            "--length", "3.0", "--interval", "4", "-oni", "frames.raw"]
    r = subprocess.run([os.path.join(BIN, "Integrate")] + args + ["--save_to", "world_hip.pcd", "--max_units", "512"],
                       cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Trajectory created from pose and segment trajectories." in r.stdout
    hip = formats.load_pcd(os.path.join(d, "world_hip.pcd"))
    hip = np.stack([hip["x"], hip["y"], hip["z"], hip["intensity"]], 1)
    assert "%d voxel points have been written." % hip.shape[0] in r.stdout
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "Integrate_ref")
    if os.path.exists(ref_bin):
        rr = subprocess.run([ref_bin] + args + ["--save_to", "world_ref.pcd"], cwd=d, capture_output=True, text=True, timeout=600,
                            env=dict(os.environ, ER_ORACLE_QUIET="1"))
        assert rr.returncode == 0, rr.stdout + rr.stderr
        ref = formats.load_pcd(os.path.join(d, "world_ref.pcd"))
        ref = np.stack([ref["x"], ref["y"], ref["z"], ref["intensity"]], 1)
    else:
        from oracle.pyoracle import OracleVolume
        seg_l, pose_l = formats.load_log(os.path.join(d, "seg.log")), formats.load_log(os.path.join(d, "pose.log"))
        from elasticreconstruction_amd import tsdf
        traj = [tsdf.mat4_mul(pose_l[f // 4].T, seg_l[f].T) for f in range(12)]
        ora = OracleVolume()
        for f in range(12):
            m = OracleVolume.reproject_matrix(traj[f], traj[0], seg_l[0].T)
            ora.Integrate(ora.Reproject(depth[f], sc["grids"][f // 4], 8, 3.0, seg_l[f].T, m), traj[f])
        ref = ora.extract_world()
    assert hip.shape == ref.shape and hip.shape[0] > 50000
    assert np.array_equal(sorted_points(hip).view(np.uint32), sorted_points(ref).view(np.uint32)), "world.pcd point sets differ"


def test_integrate_program_png_source_and_rigid_mode(gpu, tmp_path):
    """--ref_traj (rigid) + --depth_list of 16-bit PNGs + --start_from/--end_at: same volume as the raw-stream run."""
    from PIL import Image
    d = str(tmp_path)
    poses = synth.circle_trajectory(3000)[3::400][:5]
    depth = synth.to_numpy_u16(synth.render_depth(poses))
    traj = [formats.FramedTransformation(i, i, i + 1, poses[min(i, 4)]) for i in range(6)]
    formats.save_log(os.path.join(d, "traj.log"), traj)
    depth.tofile(os.path.join(d, "frames.raw"))
    with open(os.path.join(d, "list.txt"), "w") as f:
        for i in range(5):
            Image.fromarray(depth[i].reshape(480, 640)).save(os.path.join(d, "%03d.png" % i))
            f.write("%03d.png\n" % i)
    outs = []
    for src in (["-oni", "frames.raw"], ["--depth_list", "list.txt"]):
        r = subprocess.run([os.path.join(BIN, "Integrate"), "--ref_traj", "traj.log", "--start_from", "2", "--end_at", "4",
                            "--save_to", "w.pcd", "--max_units", "512"] + src, cwd=d, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "Reaching the specified end point." in r.stdout
        p = formats.load_pcd(os.path.join(d, "w.pcd"))
        outs.append(sorted_points(np.stack([p["x"], p["y"], p["z"], p["intensity"]], 1)))
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    from oracle.pyoracle import OracleVolume
    ora = OracleVolume()
    logged = formats.load_log(os.path.join(d, "traj.log"))
    for f in (2, 3, 4):
        ora.Integrate(depth[f - 1], logged[f - 1].T)
    assert np.array_equal(outs[0].view(np.uint32), sorted_points(ora.extract_world()).view(np.uint32))


def _bc_outputs(d, pairs):
    """(log, info, {pair: text}) of the last run in d; corres files are removed so the next program starts clean."""
    from corres_helpers import read_outputs
    log, info, corr = read_outputs(d, pairs)
    for k in corr:
        os.remove(d + "corres_%d_%d.txt" % k)
    return log, info, corr


def test_build_correspondence_program_equals_reference_program(gpu, tmp_path):
    """bin/BuildCorrespondence against the reference's own program on the same files and flags (two passes, see the module
    docstring), live where oracle/_ref/BuildCorrespondence_ref exists and against tests/golden/corres_golden.json always."""
    import hashlib
    from corres_helpers import REF_BIN, corres_golden, run_program, scene_digest, standard_pairs, write_refined_log, write_scene
    d = str(tmp_path) + "/"
    fr = write_scene(d)
    pairs = standard_pairs(fr, d)
    with open(d + "black.txt", "w") as f:
        f.write("4\n")
    ours = os.path.join(BIN, "BuildCorrespondence")
    g = corres_golden()
    golden_ok = scene_digest(fr) == g["scene_digest"]
    have_ref = os.path.exists(REF_BIN)
    assert golden_ok or have_ref, "neither the reference program nor a reproducible golden scene on this host"
    a1 = ["--reg_traj", d + "init.log", "--registration", "--reg_dist", "0.04", "--output_information", "--blacklist", d + "black.txt"]
    run_program(ours, a1, d)
    log, info, corr = _bc_outputs(d, pairs)
    refs = []
    if have_ref:
        run_program(REF_BIN, a1, d)
        refs.append(("live", ) + _bc_outputs(d, pairs))
    if golden_ok:
        from test_corres_golden import _parse
        gl = [formats.FramedTransformation(a, b, c, T) for a, b, c, T in _parse(g["pass1"]["log"], 4)]
        gi = [formats.FramedInformation(a, b, c, M) for a, b, c, M in _parse(g["pass1"]["info"], 6)]
        refs.append(("golden", gl, gi, None))
    for what, rlog, rinfo, rcorr in refs:
        assert [(t.id1, t.id2) for t in log] == [(t.id1, t.id2) for t in rlog], what
        assert [t.frame == -1 for t in log] == [t.frame == -1 for t in rlog] == [False, True, False, False, True, True], what
        for t, r, fi, ri in zip(log, rlog, info, rinfo):
            assert np.abs(t.T - r.T).max() <= 1e-5, (what, t.id1, t.id2, np.abs(t.T - r.T).max())
            # the two ICPs differ by ~1e-7 in T (float64 sums in another order), so a handful of borderline points may flip
            assert abs(t.frame - r.frame) <= max(3, r.frame // 1000) and fi.frame == t.frame, (what, t.frame, r.frame)
            if t.frame != -1:
                assert np.abs(fi.info - ri.info).max() <= 2e-3 * np.abs(ri.info).max(), what
    # pass 2: FindCorrespondence only, both programs start from OUR 8-decimal transforms -> exact comparisons
    write_refined_log(d + "refined.log", log, len(fr))
    a2 = ["--reg_traj", d + "refined.log", "--reg_dist", "0.04", "--output_information"]
    run_program(ours, a2, d)
    log2, info2, corr2 = _bc_outputs(d, pairs)
    assert set(corr2) == {(0, 1), (1, 2), (2, 3)}
    if have_ref:
        run_program(REF_BIN, a2, d)
        rlog2, rinfo2, rcorr2 = _bc_outputs(d, pairs)
        assert corr2 == rcorr2, "corres_<i>_<j>.txt differ from the reference program's"
        assert [(t.id1, t.id2, t.frame) for t in log2] == [(t.id1, t.id2, t.frame) for t in rlog2]
        for a, b in zip(info2, rinfo2):
            assert a.frame == b.frame and np.allclose(a.info, b.info, rtol=1e-9, atol=2e-8)      # %.8f on both sides
    if golden_ok:                                                                             # the reference's own pass 2 from ITS transforms
        with open(d + "refined.log", "w") as f:
            f.write(g["refined_log"])
        run_program(ours, a2, d)
        log3, info3, corr3 = _bc_outputs(d, pairs)
        assert {"%d_%d" % k: hashlib.sha256(v.encode()).hexdigest() for k, v in corr3.items()} == g["pass2"]["corres_sha256"]
        gl = _parse(g["pass2"]["log"], 4)
        gi = _parse(g["pass2"]["info"], 6)
        assert [(t.id1, t.id2, t.frame) for t in log3] == [(a, b, c) for a, b, c, _ in gl]
        for t, (_, _, _, T) in zip(log3, gl):
            assert np.abs(t.T - T).max() <= 1e-12                                             # same text in, same text out
        for fi, (_, _, c, M) in zip(info3, gi):
            assert fi.frame == c and np.allclose(fi.info, M, rtol=1e-9, atol=2e-8)


def test_build_correspondence_program_equals_python_mirror(gpu, tmp_path):
    d = str(tmp_path) + "/"
    frag = synth.look_at((1.5, 1.5, 1.5), (0, 0, 1)) @ np.linalg.inv(synth.basepose())
    truth = []
    for i in range(3):
        x, n = synth.sample_fragment(frag, 160000, seed=300 + i)
        P = synth.perturbation(400 + i, 1.0, 0.01) if i else np.eye(4)
        Pi = np.linalg.inv(P)
        x, n = (x @ Pi[:3, :3].T + Pi[:3, 3]).astype(np.float32), (n @ Pi[:3, :3].T).astype(np.float32)
        n[::53, 0] = np.nan
        formats.save_pcd_xyzn(d + "cloud_bin_%d.pcd" % i, x, n, binary=(i != 1))       # one ascii file on purpose
        truth.append(P)
    pairs = [formats.FramedTransformation(0, 1, 3, truth[1] @ synth.perturbation(1, 0.4, 0.004)),
             formats.FramedTransformation(0, 2, 3, synth.perturbation(2, 70, 1.2)),
             formats.FramedTransformation(1, 2, 3, np.linalg.inv(truth[1]) @ truth[2])]
    formats.save_log(d + "init.log", pairs)
    r = subprocess.run([os.path.join(BIN, "BuildCorrespondence"), "--reg_traj", d + "init.log", "--registration", "--reg_dist", "0.03",
                        "--output_information", "--save_xyzn"], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    log_c = open(d + "reg_output.log").read()
    info_c = open(d + "reg_output.info").read()
    corr_c = {k: open(d + "corres_%d_%d.txt" % k).read() for k in ((0, 1), (1, 2))}
    assert not os.path.exists(d + "corres_0_2.txt") and os.path.exists(d + "cloud_bin_xyzn_2.xyzn")
    os.makedirs(d + "py")
    app = CorresApp()
    app.out_dir = d + "py"
    app.reg_dist_, app.dist_thresh_ = 0.03, 0.015
    app.LoadData(d + "init.log", -1)
    app.output_information_ = True
    app.Registration()
    app.FindCorrespondence()
    app.Finalize()
    assert open(d + "py/reg_output.log").read() == log_c
    ic, ip = formats.load_info(d + "reg_output.info"), formats.load_info(d + "py/reg_output.info")
    assert all(np.allclose(a.info, b.info, rtol=1e-12, atol=1e-6) and a.frame == b.frame for a, b in zip(ic, ip))   # float64 atomics: order varies
    for k, txt in corr_c.items():
        assert open(d + "corres_%d_%d.txt" % k).read() == txt
    out = formats.load_log(d + "reg_output.log")
    assert [t.frame == -1 for t in out] == [False, True, False]
    assert np.abs(out[0].T - truth[1]).max() < 2e-3 and np.abs(out[2].T - np.linalg.inv(truth[1]) @ truth[2]).max() < 2e-3


def test_fragment_optimizer_program_equals_reference_program(gpu, tmp_path):
    """bin/FragmentOptimizer (GPU assembly, host regularizer + dense Cholesky) against the reference's own FragmentOptimizer
    program (oracle/_ref/FragmentOptimizer_ref: reference sources compiled in place, CHOLMOD replaced by the dense shim) on
    the same files and flags: output.ctr and pose.log of --rigid, --slac and the default non-rigid mode."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fopt_helpers import make_scene
    from test_fopt_oracle import REF_BIN, _run_ref, _write_dataset
    if not os.path.exists(REF_BIN):
        pytest.skip("oracle/_ref/FragmentOptimizer_ref is built where /root/reference exists")

    def run_ours(d, mode, extra, out):
        cmd = [os.path.join(BIN, "FragmentOptimizer")] + extra + ["--registration", os.path.join(d, "reg_output.log"), "--dir", d + "/",
                                                               "--rgbdslam", os.path.join(d, "rgbd.log"), "--interval", "1", "--blacklistpair", "0",
                                                               "--save_to", os.path.join(d, out)]
        if mode != "nonrigid":
            cmd.append("--" + mode)
        r = subprocess.run(cmd, cwd=d, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        return r.stdout

    def read_log(fn):
        return [t.T for t in formats.load_log(fn)]

    d = str(tmp_path)
    sc = make_scene(num=3, n=3000)
    _write_dataset(sc, d)
    args = ["--num", "3", "--resolution", "8", "--length", "3.0", "--iteration", "2"]
    for mode in ("rigid", "slac"):
        _run_ref(d, mode, "reg_output.log", args)
        ref_pose, ref_ctr = read_log(os.path.join(d, "pose.log")), np.loadtxt(os.path.join(d, "out_%s.ctr" % mode))
        os.rename(os.path.join(d, "pose.log"), os.path.join(d, "pose_ref_%s.log" % mode))
        out = run_ours(d, mode, args, "ours_%s.ctr" % mode)
        assert "Read " in out and "correspondences" in out
        pose, ctr = read_log(os.path.join(d, "pose.log")), np.loadtxt(os.path.join(d, "ours_%s.ctr" % mode))
        assert max(np.abs(a - b).max() for a, b in zip(pose, ref_pose)) < 1e-7, mode
        assert np.abs(ctr - ref_ctr).max() < 1e-7, mode
    sc4 = make_scene(num=3, n=3000, res=4)
    d4 = os.path.join(d, "r4")
    os.makedirs(d4)
    _write_dataset(sc4, d4)
    args4 = ["--num", "3", "--resolution", "4", "--length", "3.0", "--weight", "1.7", "--inner_iteration", "2", "--iteration", "1"]
    _run_ref(d4, "nonrigid", "reg_output.log", args4)
    ref_ctr = np.loadtxt(os.path.join(d4, "out_nonrigid.ctr"))
    run_ours(d4, "nonrigid", args4, "ours.ctr")
    assert np.abs(np.loadtxt(os.path.join(d4, "ours.ctr")) - ref_ctr).max() < 1e-6
    # --init_ctr: the non-rigid mode restarted from the lattices just written, SLAC from a jittered canonical lattice
    os.replace(os.path.join(d4, "out_nonrigid.ctr"), os.path.join(d4, "start.ctr"))
    _run_ref(d4, "nonrigid", "reg_output.log", ["--init_ctr", os.path.join(d4, "start.ctr")] + args4)
    ref_ctr = np.loadtxt(os.path.join(d4, "out_nonrigid.ctr"))
    run_ours(d4, "nonrigid", ["--init_ctr", os.path.join(d4, "start.ctr")] + args4, "ours2.ctr")
    assert np.abs(np.loadtxt(os.path.join(d4, "ours2.ctr")) - ref_ctr).max() < 1e-6
    k, j, i = np.meshgrid(np.arange(9), np.arange(9), np.arange(9), indexing="ij")
    lat = np.stack([i.ravel(), j.ravel(), k.ravel()], 1) * (3.0 / 8) + np.random.default_rng(1).normal(0, 0.002, (729, 3))
    np.savetxt(os.path.join(d, "lat.ctr"), lat, fmt="%.10f")
    _run_ref(d, "slac", "reg_output.log", ["--init_ctr", os.path.join(d, "lat.ctr")] + args)
    ref_ctr = np.loadtxt(os.path.join(d, "out_slac.ctr"))
    run_ours(d, "slac", ["--init_ctr", os.path.join(d, "lat.ctr")] + args, "ours_slac2.ctr")
    assert np.abs(np.loadtxt(os.path.join(d, "ours_slac2.ctr")) - ref_ctr).max() < 1e-7
    # --write_xyzn_sample: sample.pcd (SavePoints, OptApp.cpp:897-923; ours is binary_compressed like PCL's, the reference
    # build's stand-in writer stores the same records uncompressed) and the per-iteration lattices of the non-rigid mode
    def sample(dirname):
        c = formats.load_pcd(os.path.join(dirname, "sample.pcd"))
        return np.stack([c[k] for k in ("x", "y", "z", "normal_x", "normal_y", "normal_z", "rgb", "curvature")], 1)

    sargs = ["--write_xyzn_sample", "7", "--init_ctr", os.path.join(d, "lat.ctr")] + args
    _run_ref(d, "slac", "reg_output.log", sargs)
    ref_s = sample(d)
    os.remove(os.path.join(d, "sample.pcd"))
    run_ours(d, "slac", sargs, "ours_slac3.ctr")
    assert b"DATA binary_compressed" in open(os.path.join(d, "sample.pcd"), "rb").read(400)
    s = sample(d)
    assert s.shape == ref_s.shape and len(s) > 300 and np.abs(s - ref_s).max() < 2e-6
    _run_ref(d4, "nonrigid", "reg_output.log", ["--write_xyzn_sample", "5"] + args4)
    ref_s = sample(d4)
    dumps = ["itr0.ctr", "itr0_inner0_out.ctr", "itr0_inner1_out.ctr"]
    ref_d = [np.loadtxt(os.path.join(d4, f)) for f in dumps]
    for f in dumps + ["sample.pcd"]:
        os.remove(os.path.join(d4, f))
    run_ours(d4, "nonrigid", ["--write_xyzn_sample", "5"] + args4, "ours3.ctr")
    s = sample(d4)
    assert s.shape == ref_s.shape and np.abs(s - ref_s).max() < 5e-6
    for f, r in zip(dumps, ref_d):
        assert np.abs(np.loadtxt(os.path.join(d4, f)) - r).max() < 1e-6, f
    # beyond --dense_limit unknowns the program keeps and factors the system as the block-sparse lower triangle of fragment
    # blocks (the reference: CHOLMOD's sparse Cholesky, OptApp.cpp:155-211): same files as the dense run and as the reference
    out = run_ours(d4, "nonrigid", ["--dense_limit", "100"] + args4, "ours_blocked.ctr")
    assert "block-sparse Cholesky over 3 fragment blocks" in out
    blocked = np.loadtxt(os.path.join(d4, "ours_blocked.ctr"))
    assert np.abs(blocked - np.loadtxt(os.path.join(d4, "ours.ctr"))).max() < 1e-9
    _run_ref(d4, "nonrigid", "reg_output.log", args4)
    assert np.abs(blocked - np.loadtxt(os.path.join(d4, "out_nonrigid.ctr"))).max() < 1e-6


def _same_world(a, b):
    """SaveWorld keeps |sdf| < 0.98: a voxel whose sdf sits at the threshold may flip under the 1e-5 re-rounding of a merged unit; everything else must be
    the same voxel with an intensity within 1e-5."""
    if a.shape == b.shape:
        assert np.array_equal(a[:, :3], b[:, :3]) and np.abs(a[:, 3] - b[:, 3]).max() <= 1e-5
    else:
        assert abs(a.shape[0] - b.shape[0]) <= 1e-4 * a.shape[0]


def integrate_multi_gpu_case(d, gpus_args, same_device):
    """bin/Integrate with `gpus_args` (+ --same_device on a one-GPU box) in both shard modes against the single-GPU program (also called by
    tests/test_distributed_gpu.py with real devices):
       * --shard frame (default): frame blocks per worker, er_tsdf_allreduce leaves the merged volume DISTRIBUTED by unit owner (round 6), world.pcd is
         assembled from the owners' extractions in ascending key order: the same points within 1e-5;
       * --merge_root 0: the same merge gathered on the first GPU before SaveWorld; ER_MERGE_IMPL=ring: round 5's protocol -- all the same points;
       * --shard unit: every worker is fed every frame and owns a third of the units, no collective: world.pcd BYTE-identical to the single-GPU one."""
    sc = synth.make_scenario(12, interval=4, warp=True, amplitude=0.004, seed=21)
    depth = synth.to_numpy_u16(sc["depth"])
    write_integrate_inputs(d, sc, depth)
    args = ["--pose_traj", "pose.log", "--seg_traj", "seg.log", "--ctr", "grids.ctr", "--num", "3", "--resolution", "8",
            "--length", "3.0", "--interval", "4", "-oni", "frames.raw", "--max_units", "512"]
    sd = ["--same_device"] if same_device else []

    def run(extra, out, env=None):
        r = subprocess.run([os.path.join(BIN, "Integrate")] + args + extra + ["--save_to", out], cwd=d, capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, **(env or {})))
        assert r.returncode == 0, r.stdout + r.stderr
        p = formats.load_pcd(os.path.join(d, out))
        return np.stack([p["x"], p["y"], p["z"], p["intensity"]], 1), r
    one, _ = run([], "w1.pcd")
    a = sorted_points(one)
    where = "the loopback transport" if same_device else "RCCL"
    n = gpus_args[1]
    looped, r = run(gpus_args + sd, "wl.pcd")
    assert "merged %s GPU volumes over %s" % (n, where) in r.stderr and "left distributed" in r.stderr, r.stderr
    _same_world(a, sorted_points(looped))
    # world.pcd of the distributed result lists the units in ascending key order like the single-GPU program: same order of the points up to the
    # threshold flips
    if looped.shape == one.shape:
        assert np.array_equal(looped[:, :3], one[:, :3])
    rooted, r = run(gpus_args + sd + ["--merge_root", "0"], "wr.pcd")
    assert "gathered on GPU 0" in r.stderr, r.stderr
    _same_world(a, sorted_points(rooted))
    assert np.array_equal(sorted_points(rooted).view(np.uint32), sorted_points(looped).view(np.uint32)), "gathered and distributed results differ"
    ring, r = run(gpus_args + sd + ["--merge_root", "0"], "wg.pcd", env={"ER_MERGE_IMPL": "ring"})
    _same_world(a, sorted_points(ring))
    sharded, _ = run(gpus_args + ["--shard", "unit"] + sd, "wu.pcd")
    assert np.array_equal(one.view(np.uint32), sharded.view(np.uint32)), "unit-shard world.pcd differs from the single-GPU one"
    with open(os.path.join(d, "w1.pcd"), "rb") as f1, open(os.path.join(d, "wu.pcd"), "rb") as f2:
        assert f1.read() == f2.read()
    return one, run


def test_integrate_program_multi_gpu_modes(gpu, tmp_path):
    """bin/Integrate --gpus on the one GPU of this box: --gpus 3 --same_device in every mode (integrate_multi_gpu_case), and
    --gpus 1 --force_merge: er_tsdf_allreduce on a one-rank RCCL communicator must leave the volume as it was."""
    one, run = integrate_multi_gpu_case(str(tmp_path), ["--gpus", "3"], same_device=True)
    merged, r = run(["--gpus", "1", "--force_merge"], "wm.pcd")
    assert "merged 1 GPU volumes over RCCL" in r.stderr
    assert np.array_equal(sorted_points(one).view(np.uint32), sorted_points(merged).view(np.uint32))     # one rank: nothing is summed, nothing moves


def build_correspondence_multi_gpu_case(d, gpus):
    """bin/BuildCorrespondence --gpus N (pair p -> GPU p mod N, no collective) writes the same files as with one GPU."""
    from corres_helpers import run_program, standard_pairs, write_scene
    d = d + "/"
    fr = write_scene(d)
    pairs = standard_pairs(fr, d)
    ours = os.path.join(BIN, "BuildCorrespondence")
    a1 = ["--reg_traj", d + "init.log", "--registration", "--reg_dist", "0.04", "--output_information"]
    run_program(ours, a1, d)
    log1, info1, corr1 = _bc_outputs(d, pairs)
    txt1 = open(d + "reg_output.log").read()
    run_program(ours, a1 + ["--gpus", str(gpus)], d)
    log2, info2, corr2 = _bc_outputs(d, pairs)
    assert open(d + "reg_output.log").read() == txt1 and corr1 == corr2
    for x, y in zip(info1, info2):
        assert x.frame == y.frame and np.array_equal(x.info, y.info)


def test_abi_allreduce_on_a_one_rank_communicator_is_the_identity(gpu):
    """er_tsdf_allreduce through RCCL with ONE rank (RCCL refuses two ranks on one GPU, and this box has one): export ->
    all-reduce -> import must be the identity up to re-rounding, and er_frame_block must tile.  The protocol with world = 2 / 3
    runs on the CPU: tests/test_distributed_cpu.py::test_c_merge_protocol_world_1_2_3_on_threads (the same header)."""
    import ctypes as C
    from elasticreconstruction_amd import _ffi, parallel
    from elasticreconstruction_amd.tsdf import TSDFVolume
    sc = synth.make_scenario(8, interval=4, warp=False)
    depth = synth.to_numpy_u16(sc["depth"])
    vol = TSDFVolume(max_units=256)
    vol.IntegrateFrames(depth, sc["traj"])
    before = {int(k): vol.read_unit(k) for k in vol.unit_keys()}
    L = _ffi.lib()
    dev = (C.c_int * 1)(0)
    comm = (C.c_void_p * 1)()
    _ffi.check(L.er_comm_create_local(1, dev, comm), "er_comm_create_local")
    nu = C.c_int(0)
    _ffi.check(L.er_tsdf_allreduce(vol._h, comm[0], -1, C.byref(nu)), "er_tsdf_allreduce")
    assert nu.value == len(before) and L.er_comm_world(comm[0]) == 1 and L.er_comm_rank(comm[0]) == 0
    for k, (s0, w0) in before.items():
        s1, w1 = vol.read_unit(k)
        assert np.array_equal(w0, w1) and np.array_equal(s1.view(np.uint32), s0.view(np.uint32))    # one rank: every unit is its own, nothing is summed
    L.er_comm_destroy(comm[0])
    vol.close()
    for n, w in ((10000, 8), (3000, 4), (7, 3), (2, 4)):
        lo, hi = C.c_int(), C.c_int()
        blocks = []
        for r in range(w):
            L.er_frame_block(n, r, w, C.byref(lo), C.byref(hi))
            blocks.append((lo.value, hi.value))
            assert (lo.value, hi.value) == parallel.frame_block(n, r, w)
        assert blocks[0][0] == 0 and blocks[-1][1] == n


def test_chain_on_own_outputs_from_depth_frames_to_the_warped_volume(gpu, tmp_path):
    """every stage consumes what the stage before it wrote (README.txt: Modules; the pipeline's order is
    fragments -> BuildCorrespondence -> FragmentOptimizer -> Integrate) -- nothing is handed over in memory:
      1. four 50-frame sweeps of depth images -> one TSDF sub-volume each through the library's Integrate path -> zero crossings + gradient normals
         -> cloud_bin_<i>.pcd WITH their NaN normals (synth.kinfu_fragment: what pcl_kinfu's extraction leaves);
      2. bin/BuildCorrespondence --registration --save_xyzn on those files + an initial pair log -> reg_output.log, corres_<i>_<j>.txt, cloud_bin_xyzn_<i>.xyzn;
      3. bin/FragmentOptimizer --slac on the files of (2) + noisy initial poses -> pose.log, output.ctr;
      4. bin/Integrate --pose_traj pose.log --ctr output.ctr on the ORIGINAL 200 depth frames -> world.pcd.
    Checked: every program's input files are accepted by the corresponding reference program (oracle/_ref/*_ref, the reference sources compiled in place)
    with the same results -- transforms 1e-5, correspondence files after a FindCorrespondence-only pass byte for byte, poses and lattices 1e-6, the final
    voxel list as a bit-identical point set -- and the end result is RIGHT: the optimised poses are closer to the ground truth than the initial ones and the
    near-zero voxels of the final volume lie within 1.5 voxels of the analytic room and sphere."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from corres_helpers import REF_BIN as BC_REF, read_outputs, run_program, write_refined_log
    from test_fopt_oracle import REF_BIN as FO_REF
    d = str(tmp_path) + "/"
    NUM, FR, TOTAL = 4, 50, 50
    base = synth.basepose(3.0)
    # ---- 1. depth frames -> sub-volumes -> fragments on disk --------------------------------------------------------------------------------
    F, seg, depth = [], [], []
    for i in range(NUM):
        x, n, Fi, st = synth.kinfu_fragment(i, TOTAL, 60000)
        assert st["nan_normals"] > 0 and len(x) == 60000
        formats.save_pcd_xyzn(d + "cloud_bin_%d.pcd" % i, x, n, binary=True)
        W = synth.kinfu_camera_path(i, TOTAL, FR)
        F.append(Fi)
        seg.extend(np.linalg.inv(Fi) @ W[j] for j in range(FR))
        depth.append(synth.to_numpy_u16(synth.render_depth(W, device="cuda:0")))
    depth = np.concatenate(depth)
    gt = [np.linalg.inv(F[0]) @ F[i] for i in range(NUM)]                       # fragment poses relative to fragment 0
    # ---- 2. BuildCorrespondence ---------------------------------------------------------------------------------------------------------------
    pairs = [formats.FramedTransformation(i, j, NUM, np.linalg.inv(gt[i]) @ gt[j] @ synth.perturbation(31 * i + j, 1.0, 0.01))
             for (i, j) in [(0, 1), (1, 2), (2, 3), (0, 2), (1, 3)]]
    formats.save_log(d + "init.log", pairs)
    a1 = ["--reg_traj", d + "init.log", "--registration", "--save_xyzn", "--output_information"]
    run_program(os.path.join(BIN, "BuildCorrespondence"), a1, d)
    log, info, corr = read_outputs(d, pairs)
    assert all(t.frame > 10000 for t in log) and len(corr) == len(pairs), [t.frame for t in log]
    for t in log:                                                                 # the ICP pulled every pair onto the ground truth (within a voxel)
        assert np.abs(t.T - np.linalg.inv(gt[t.id1]) @ gt[t.id2]).max() < 6e-3, (t.id1, t.id2)
    xyzn = {i: open(d + "cloud_bin_xyzn_%d.xyzn" % i).read() for i in range(NUM)}
    if os.path.exists(BC_REF):                                                    # the reference program on OUR .pcd files and init.log
        for k in corr:
            os.rename(d + "corres_%d_%d.txt" % k, d + "ours_corres_%d_%d.txt" % k)
        run_program(BC_REF, a1, d)
        rlog, rinfo, rcorr = read_outputs(d, pairs)
        assert {i: open(d + "cloud_bin_xyzn_%d.xyzn" % i).read() for i in range(NUM)} == xyzn, "cloud_bin_xyzn_<i>.xyzn differ from the reference program's"
        for t, r in zip(log, rlog):
            assert (t.id1, t.id2) == (r.id1, r.id2) and np.abs(t.T - r.T).max() <= 1e-5 and abs(t.frame - r.frame) <= max(3, r.frame // 1000)
        # FindCorrespondence-only from OUR 8-decimal transforms: byte-identical files
        write_refined_log(d + "refined.log", log, NUM)
        a2 = ["--reg_traj", d + "refined.log", "--output_information"]
        for prog, tag in ((os.path.join(BIN, "BuildCorrespondence"), "ours"), (BC_REF, "ref")):
            run_program(prog, a2, d)
            out = read_outputs(d, pairs)
            if tag == "ours":
                log2, corr2 = out[0], out[2]
            else:
                assert corr2 == out[2] and [(t.id1, t.id2, t.frame) for t in log2] == [(t.id1, t.id2, t.frame) for t in out[0]]
        os.replace(d + "reg_output.log", d + "reg_pass2.log")
        for k in corr:
            os.replace(d + "ours_corres_%d_%d.txt" % k, d + "corres_%d_%d.txt" % k)   # stage 3 reads the registration pass's own files
        formats.save_log(d + "reg_output.log", log)
    # ---- 3. FragmentOptimizer --slac ----------------------------------------------------------------------------------------------------------
    for i in range(NUM):                                                          # stage 3 reads what stage 2 WROTE: cloud_bin_xyzn_<i>.xyzn (OptApp.cpp:86-96 prefers a
        os.replace(d + "cloud_bin_%d.pcd" % i, d + "stage1_cloud_bin_%d.pcd" % i)   # cloud_bin_<i>.pcd when one is there; the stand-in PCL of the reference build only reads PointXYZRGBNormal files)
    init = [gt[i] @ synth.perturbation(90 + i, 0.4, 0.006) if i else gt[i] for i in range(NUM)]
    with open(d + "rgbd.log", "w") as fh:                                         # InitIPose, OptApp.cpp:49-72: ipose = basepose * traj[0]^-1 * traj[i] * basepose^-1
        for f, P in enumerate(init):
            g = np.linalg.inv(base) @ P @ base
            fh.write("%d\t%d\t%d\n" % (f, f, f + 1))
            fh.write("\n".join("%.8f %.8f %.8f %.8f" % tuple(r) for r in g) + "\n")
    a3 = ["--slac", "--registration", d + "reg_output.log", "--dir", d, "--rgbdslam", d + "rgbd.log", "--interval", "1", "--num", str(NUM), "--resolution", "8",
          "--length", "3.0", "--iteration", "4", "--blacklistpair", "0"]
    run_program(os.path.join(BIN, "FragmentOptimizer"), a3 + ["--save_to", d + "output.ctr"], d)
    pose = [t.T for t in formats.load_log(d + "pose.log")]
    ctr = np.loadtxt(d + "output.ctr")
    assert len(pose) == NUM and ctr.shape == (NUM * 729, 3)
    err0 = max(np.abs(init[i] - gt[i]).max() for i in range(NUM))
    err1 = max(np.abs(pose[i] - pose[0] @ gt[i]).max() for i in range(NUM))      # (the gauge: everything relative to the optimised pose of fragment 0)
    assert err1 < 0.5 * err0 and err1 < 4e-3, "SLAC did not pull the poses onto the ground truth: %.3g -> %.3g" % (err0, err1)
    if os.path.exists(FO_REF):                                                    # the reference program on OUR reg_output.log / corres / xyzn files
        os.replace(d + "pose.log", d + "pose_ours.log")
        run_program(FO_REF, a3 + ["--save_to", d + "output_ref.ctr"], d)
        rpose = [t.T for t in formats.load_log(d + "pose.log")]
        assert max(np.abs(a - b).max() for a, b in zip(pose, rpose)) < 1e-6 and np.abs(ctr - np.loadtxt(d + "output_ref.ctr")).max() < 1e-6
        os.replace(d + "pose_ours.log", d + "pose.log")
    # ---- 4. Integrate with the optimised poses and lattices -------------------------------------------------------------------------------------
    plog = formats.load_log(d + "pose.log")
    formats.save_log(d + "pose5.log", plog + [formats.FramedTransformation(NUM, NUM, NUM + 1, plog[-1].T)])   # F + 1 entries: all F frames are integrated
    n = NUM * FR
    formats.save_log(d + "seg.log", [formats.FramedTransformation(f, f, f + 1, seg[f]) for f in range(n)] +
                     [formats.FramedTransformation(n + j, n + j, n + j + 1, seg[-1]) for j in range(FR)])
    depth.tofile(d + "frames.raw")
    a4 = ["--pose_traj", "pose5.log", "--seg_traj", "seg.log", "--ctr", "output.ctr", "--num", str(NUM), "--resolution", "8", "--length", "3.0",
          "--interval", str(FR), "-oni", "frames.raw"]
    r = run_program(os.path.join(BIN, "Integrate"), a4 + ["--save_to", "world.pcd", "--max_units", "1024"], d)
    w = formats.load_pcd(d + "world.pcd")
    w = np.stack([w["x"], w["y"], w["z"], w["intensity"]], 1)
    assert "%d voxel points have been written." % w.shape[0] in r.stdout and w.shape[0] > 200000
    ref_int = os.path.join(ROOT, "oracle", "_ref", "Integrate_ref")
    if os.path.exists(ref_int):                                                   # the reference program on OUR pose.log / output.ctr: the same voxels, bit for bit
        run_program(ref_int, a4 + ["--save_to", "world_ref.pcd"], d, timeout=900)
        wr = formats.load_pcd(d + "world_ref.pcd")
        wr = np.stack([wr["x"], wr["y"], wr["z"], wr["intensity"]], 1)
        assert np.array_equal(sorted_points(w).view(np.uint32), sorted_points(wr).view(np.uint32)), "world.pcd differs from the reference program's"
    # the surface of the final volume is where the scene is: voxels next to the zero crossing (|tsdf| < 0.15 = 4.5 mm of the 30 mm band), lifted from the
    # volume's frame (= fragment 0's cube, up to the optimised pose of fragment 0) into the room
    near = w[np.abs(w[:, 3]) < 0.15]
    p = near[:, :3].astype(np.float64) * (3.0 / 512.0)
    M = F[0] @ np.linalg.inv(pose[0])
    q = p @ M[:3, :3].T + M[:3, 3]
    dw = np.minimum(np.abs(q - synth.ROOM_LO), np.abs(q - synth.ROOM_HI)).min(axis=1)
    ds = np.abs(np.linalg.norm(q - np.asarray(synth.SPHERE_C), axis=1) - synth.SPHERE_R)
    dist = np.minimum(dw, ds)
    vox = 3.0 / 512.0
    assert len(near) > 20000 and np.mean(dist < 1.5 * vox) > 0.97 and np.percentile(dist, 99.5) < 3.0 * vox, \
        "final surface off the scene: %.3f within 1.5 voxels, 99.5 %% at %.2f voxels" % (np.mean(dist < 1.5 * vox), np.percentile(dist, 99.5) / vox)
    print("chain: %d correspondences over %d pairs, pose error %.2g -> %.2g, %d near-zero voxels, %.1f %% within 1.5 voxels of the scene (median %.2f voxels)"
          % (sum(t.frame for t in log), len(log), err0, err1, len(near), 100 * np.mean(dist < 1.5 * vox), np.median(dist) / vox))
