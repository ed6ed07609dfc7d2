"""Parity tests proper for path A: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs, against the committed golden digests of the reference build, and -- at full config
sizes -- through size-independent properties.  Bar: BIT-EXACT sdf_/weight_ (float32 bit patterns),
identical unit key sets; only the multi-GPU frame-split merge is a float-tolerance test (1e-5,
SURVEY.md 8e: it changes the summation order by construction)."""
import os

import numpy as np
import pytest

import helpers
from elasticreconstruction_amd import _ffi, synth
from elasticreconstruction_amd.tsdf import IntegrateApp, TSDFVolume
from oracle.pyoracle import OracleVolume

pytestmark = pytest.mark.gpu


def test_scale_depth_bit_exact(gpu):
    poses, depth = helpers.golden_rigid()
    vol, ora = TSDFVolume(max_units=8), OracleVolume()
    for f in (0, 3):
        assert np.array_equal(vol.ScaleDepth(depth[f]).view(np.uint32), ora.ScaleDepth(depth[f]).view(np.uint32))
    assert helpers.digest(vol.ScaleDepth(depth[0])) == helpers.golden()["scale_depth_frame0"]
    # edge cases: all zero, all max, ragged values
    for d in (np.zeros(307200, np.uint16), np.full(307200, 65535, np.uint16),
              np.random.RandomState(1).randint(0, 65536, 307200).astype(np.uint16)):
        assert np.array_equal(vol.ScaleDepth(d).view(np.uint32), ora.ScaleDepth(d).view(np.uint32))
    vol.close()


def test_integrate_golden_rigid_frame_by_frame(gpu):
    """One er_tsdf_integrate call per frame == reference digests == oracle."""
    poses, depth = helpers.golden_rigid()
    g = helpers.golden()["rigid"]
    vol, ora = TSDFVolume(max_units=256), OracleVolume()
    for i in range(len(poses)):
        vol.Integrate(depth[i], poses[i])
        ora.Integrate(depth[i], poses[i])
    helpers.assert_volumes_identical(vol, ora, "rigid")
    d = helpers.volume_digest(vol)
    assert d["keys"] == g["keys"] and d["sha256"] == g["sha256"] and d["sum_weight"] == g["sum_weight"]
    assert vol.sum_weight() == g["sum_weight"]
    vol.close()


def test_integrate_golden_rigid_batched_equals_sequential(gpu):
    """The fused batch (each voxel loaded once, frames applied in order) is bit-identical."""
    poses, depth = helpers.golden_rigid()
    g = helpers.golden()["rigid"]
    vol = TSDFVolume(max_units=256)
    vol.IntegrateFrames(depth, poses)
    d = helpers.volume_digest(vol)
    assert d["keys"] == g["keys"] and d["sha256"] == g["sha256"]
    vol.close()


def test_integrate_golden_warp(gpu):
    sc = helpers.golden_warp()
    g = helpers.golden()["warp"]
    depth = synth.to_numpy_u16(sc["depth"])
    warp = synth.warp_arrays(sc)
    vol = TSDFVolume(max_units=256)
    for f in range(sc["n"]):
        d = vol.Reproject(depth[f], sc["grids"][warp["grid_index"][f]], sc["resolution"], sc["length"], warp["seg"][f], warp["madj"][f])
        assert helpers.digest(d) == g["reprojected_depth"][f], "re-projected depth of frame %d" % f
    vol.IntegrateFrames(depth, sc["traj"], warp)
    d = helpers.volume_digest(vol)
    assert d["keys"] == g["keys"] and d["sha256"] == g["sha256"]
    vol.close()


def test_batch_boundary_and_device_resident_input(gpu):
    """70 frames (> ER_MAX_BATCH = 64, so two fused launches) from HBM-resident depth, moving pose, warp on;
    identical pose repeated inside the batch as well (same unit hit by consecutive frames)."""
    import torch
    sc = synth.make_scenario(70, interval=35, warp=True, amplitude=0.004, device="cuda:0", revolutions=0.08)
    depth_dev = sc["depth"]
    depth = synth.to_numpy_u16(depth_dev)
    warp = synth.warp_arrays(sc)
    vol, ora = TSDFVolume(max_units=512), OracleVolume()
    vol.IntegrateFrames(None, sc["traj"], warp, device_ptr=depth_dev.data_ptr())
    torch.cuda.synchronize()
    helpers.oracle_run(ora, sc, depth, warp)
    n = helpers.assert_volumes_identical(vol, ora, "70-frame warp")
    assert n > 20
    assert vol.sum_weight() == ora.sum_weight()
    # SaveWorld: same point set (the reference's unordered_map order is not canonical)
    wg, wo = vol.extract_world(), ora.extract_world()
    assert wg.shape == wo.shape and wg.shape[0] > 10000
    assert np.array_equal(wg.view(np.uint32), wo.view(np.uint32)), "SaveWorld point lists differ"
    vol.close()


def test_reproject_zero_depth_reset_semantics(gpu):
    """A re-projected depth of 0 RESETS a z-buffer cell in the reference's sequential loop
    (IntegrateApp.cpp:260-263); the device replays such cells exactly.  Contrived input: a grid that
    collapses everything onto the camera centre region so many pixels land within 0.5 mm."""
    res, length = 2, 3.0
    n1 = res + 1
    k, j, i = np.meshgrid(np.arange(n1), np.arange(n1), np.arange(n1), indexing="ij")
    verts = np.stack([i.ravel(), j.ravel(), k.ravel()], 1).astype(np.float64) * (length / res)
    rng = np.random.RandomState(5)
    depth = rng.randint(400, 3000, 307200).astype(np.uint16)
    depth[rng.rand(307200) < 0.1] = 0
    seg = synth.basepose(length)
    madj = np.linalg.inv(seg)
    # squash z so that warped points sit between 0.1 mm and 1.4 mm in front of the camera
    ctr = verts.copy()
    ctr[:, 2] = -0.3 + 0.0001 + (verts[:, 2] / length) * 0.0013
    ctr[:, 0] = 1.5 + (verts[:, 0] - 1.5) * 0.0005
    ctr[:, 1] = 1.5 + (verts[:, 1] - 1.5) * 0.0005
    ctr = ctr.astype(np.float32)
    vol, ora = TSDFVolume(max_units=8), OracleVolume()
    dg = vol.Reproject(depth, ctr, res, length, seg, madj)
    do = ora.Reproject(depth, ctr, res, length, seg, madj)
    assert (do == 0).sum() < do.size and (do == 1).sum() > 0, "test input does not exercise the 0/1 mm boundary"
    assert np.array_equal(dg, do), "%d pixels differ" % int((dg != do).sum())
    # and the flag is re-armed: a normal frame afterwards is still exact
    sc = helpers.golden_warp()
    d0 = synth.to_numpy_u16(sc["depth"])[0]
    w = synth.warp_arrays(sc)
    assert np.array_equal(vol.Reproject(d0, sc["grids"][0], sc["resolution"], sc["length"], w["seg"][0], w["madj"][0]),
                          ora.Reproject(d0, sc["grids"][0], sc["resolution"], sc["length"], w["seg"][0], w["madj"][0]))
    vol.close()


def test_reproject_zero_depth_reset_inside_a_batch(gpu):
    """The same order-dependent case inside er_tsdf_integrate_frames (the pipeline's own consumer of the z-buffer, k_prepare): a
    40-frame batch in which frames 3, 20 and 37 (bit 5 of the SECOND flag word) go through a grid that collapses the points onto
    the camera (zero writes, flagged, replayed by k_reproject_fix) and all the others through a mild grid.  Every frame's
    re-projected depth decides which units it touches and what it integrates: the volume must equal the oracle's frame-by-frame
    Reproject + Integrate bit for bit -- twice, so that the second batch meets re-armed replay buffers on both pre-pass streams."""
    res, length = 2, 3.0
    n1 = res + 1
    k, j, i = np.meshgrid(np.arange(n1), np.arange(n1), np.arange(n1), indexing="ij")
    verts = np.stack([i.ravel(), j.ravel(), k.ravel()], 1).astype(np.float64) * (length / res)
    squash = verts.copy()
    squash[:, 2] = -0.3 + 0.0001 + (verts[:, 2] / length) * 0.0013
    squash[:, 0] = 1.5 + (verts[:, 0] - 1.5) * 0.0005
    squash[:, 1] = 1.5 + (verts[:, 1] - 1.5) * 0.0005
    mild = verts + np.random.RandomState(3).normal(0, 0.004, verts.shape)
    grids = np.stack([mild, squash]).astype(np.float32)
    n = 40
    special = (3, 20, 37)
    rng = np.random.RandomState(6)
    traj = synth.circle_trajectory(3000)[::70][:n]
    depth = synth.to_numpy_u16(synth.render_depth(traj)).copy()        # the room as the cameras see it (a few dozen units per frame)
    depth[rng.rand(n, 307200) < 0.05] = 0
    seg = np.stack([synth.basepose(length)] * n)
    madj = np.stack([np.linalg.inv(seg[0])] * n)
    warp = dict(ctr=grids, resolution=res, length=np.float32(length), grid_index=np.array([1 if f in special else 0 for f in range(n)], np.int32),
                seg=seg, madj=madj)
    vol, ora = TSDFVolume(max_units=2048), OracleVolume()
    zero_cells = 0
    for rep in range(2):
        vol.IntegrateFrames(depth, traj, warp)
        for f in range(n):
            d = ora.Reproject(depth[f], grids[warp["grid_index"][f]], res, length, seg[f], madj[f])
            if f in special:
                zero_cells += int((d == 1).sum())
            ora.Integrate(d, traj[f])
        helpers.assert_volumes_identical(vol, ora, "zero-write replay inside a batch, pass %d" % rep)
    assert zero_cells > 0, "test input does not exercise the 0/1 mm boundary"
    assert vol.sum_weight() == ora.sum_weight() and vol.sum_weight() > 1e6
    vol.close()


def test_edge_cases_empty_and_far_frames(gpu):
    """Empty depth frame (no unit touched), frame entirely beyond integration_trunc (units allocated but
    all-zero), custom camera file."""
    cam = np.array([517.3, 516.5, 318.6, 255.3, 2.5, 1.2], np.float32)
    poses = synth.circle_trajectory(3000)[7::500][:3]
    depth = synth.to_numpy_u16(synth.render_depth(poses, cam=tuple(cam[:4])))
    depth[1] = 0                      # empty frame in the middle of a batch
    vol, ora = TSDFVolume(camera=cam, max_units=256), OracleVolume(camera=cam)
    vol.IntegrateFrames(depth, poses)
    for i in range(3):
        ora.Integrate(depth[i], poses[i])
    helpers.assert_volumes_identical(vol, ora, "trunc 1.2 m / empty frame")
    keys = vol.unit_keys()
    zero_units = sum(1 for k in keys if not vol.read_unit(k)[1].any())
    assert zero_units > 0, "expected far units that are allocated but never updated"
    vol.close()


def test_pool_overflow_is_reported(gpu):
    poses, depth = helpers.golden_rigid()
    vol = TSDFVolume(max_units=4)
    vol.IntegrateFrames(depth[:1], poses[:1])
    with pytest.raises(_ffi.ErError, match="pool exhausted"):
        vol.unit_count()
    vol.close()


def test_integrate_app_gating_matches_reference_flow(gpu, tmp_path):
    """IntegrateApp mirrors CIntegrateApp::Init/Execute: file inputs, 1-based frame ids, frame_ == -1 skip,
    start_from/end_at window, end-of-trajectory off-by-one, Reproject's frame_id > interval*num exit."""
    import os
    from elasticreconstruction_amd import formats, tsdf
    sc = synth.make_scenario(8, interval=4, warp=True, amplitude=0.003, seed=3)
    depth = synth.to_numpy_u16(sc["depth"])
    d = str(tmp_path)
    pose = [formats.FramedTransformation(i, i, i + 1, sc["pose"][i]) for i in range(2)]
    seg = [formats.FramedTransformation(i, i, i + 1, sc["seg"][i]) for i in range(8)]
    seg[2].frame = 3
    formats.save_log(os.path.join(d, "pose.log"), pose)
    formats.save_log(os.path.join(d, "seg.log"), seg)
    formats.save_ctr(os.path.join(d, "g.ctr"), sc["grids"])
    app = IntegrateApp(max_units=256)
    app.pose_filename_, app.seg_filename_, app.ctr_filename_ = os.path.join(d, "pose.log"), os.path.join(d, "seg.log"), os.path.join(d, "g.ctr")
    app.ctr_num_, app.ctr_interval_ = 2, 4
    app.start_from_, app.end_at_ = 2, 7
    app.pcd_filename_ = os.path.join(d, "world.pcd")
    app.Init()
    assert len(app.traj_) == 8
    for f in range(1, 9):
        if app.exit_:
            break
        app.Execute(f, depth[f - 1])
    n_pts = app.Finish()
    # reference flow: frames 2..7 integrated (1 skipped by start_from, 8 hits frame_id >= traj size -> exit)
    assert app.frames_integrated == 6 and app.exit_
    seg_l, pose_l = formats.load_log(os.path.join(d, "seg.log")), formats.load_log(os.path.join(d, "pose.log"))
    traj = [tsdf.mat4_mul(pose_l[f // 4].T, seg_l[f].T) for f in range(8)]
    ora = OracleVolume()
    for f in range(1, 7):
        m = OracleVolume.reproject_matrix(traj[f], traj[0], seg_l[0].T)
        dd = ora.Reproject(depth[f], sc["grids"][f // 4], 8, 3.0, seg_l[f].T, m)
        ora.Integrate(dd, traj[f])
    helpers.assert_volumes_identical(app.volume_, ora, "IntegrateApp")
    pcd = formats.load_pcd(os.path.join(d, "world.pcd"))
    assert len(pcd["x"]) == n_pts == ora.extract_world().shape[0]


def test_frame_split_merge_two_virtual_ranks(gpu):
    """SURVEY.md 8e on one GPU: two volumes integrate disjoint contiguous frame blocks, export sdf*w / w,
    the planes are summed (what the RCCL all-reduce does), a third volume imports.  Weights must be exact,
    tsdf within 1e-5 of the single-volume result (summation order differs by construction)."""
    import torch
    sc = synth.make_scenario(12, interval=6, warp=False, device="cuda:0", revolutions=0.05)
    depth = synth.to_numpy_u16(sc["depth"])
    full = TSDFVolume(max_units=256)
    full.IntegrateFrames(depth, sc["traj"])
    parts = [TSDFVolume(max_units=256), TSDFVolume(max_units=256)]
    parts[0].IntegrateFrames(depth[:6], sc["traj"][:6])
    parts[1].IntegrateFrames(depth[6:], sc["traj"][6:])
    keys = np.union1d(parts[0].unit_keys(), parts[1].unit_keys()).astype(np.int32)
    assert np.array_equal(keys, full.unit_keys())
    bufs = [torch.empty((len(keys), 2, 64 ** 3), dtype=torch.float32, device="cuda:0") for _ in range(2)]
    for p, b in zip(parts, bufs):
        p.export_weighted(keys, b.data_ptr())
        p.synchronize()
    total = bufs[0] + bufs[1]
    merged = TSDFVolume(max_units=256)
    merged.import_weighted(keys, total.data_ptr())
    merged.synchronize()
    worst = 0.0
    for k in keys:
        sf, wf = full.read_unit(k)
        sm, wm = merged.read_unit(k)
        assert np.array_equal(wf, wm), "merged weights differ in unit %d" % k
        worst = max(worst, float(np.abs(sf - sm).max()))
    assert worst <= 1e-5, "merged tsdf differs by %.3g" % worst
    # Round 5, the sparse merge (csrc/er_merge_protocol.h) with rank 0 as the root: only the units BOTH blocks touched go through the sum; a unit only
    # rank 1 touched travels raw (er_tsdf_export_raw -> er_tsdf_import_raw) and must arrive BIT FOR BIT -- and equal the single-volume result exactly,
    # since no frame of rank 0 ever reached it; the root's own single-toucher units do not move at all.
    k0, k1 = parts[0].unit_keys(), parts[1].unit_keys()
    multi = np.intersect1d(k0, k1).astype(np.int32)
    only1 = np.setdiff1d(k1, k0).astype(np.int32)
    only0 = np.setdiff1d(k0, k1).astype(np.int32)
    assert len(multi) > 0 and len(only1) > 0 and len(only0) > 0, (len(multi), len(only0), len(only1))
    mb = [torch.empty((len(multi), 2, 64 ** 3), dtype=torch.float32, device="cuda:0") for _ in range(2)]
    for p, b in zip(parts, mb):
        p.export_weighted(multi, b.data_ptr())
        p.synchronize()
    raw = torch.empty((len(only1), 2, 64 ** 3), dtype=torch.float32, device="cuda:0")
    parts[1].export_raw(only1, raw.data_ptr())
    parts[1].synchronize()
    msum = mb[0] + mb[1]
    root = parts[0]
    root.import_weighted(multi, msum.data_ptr())
    root.import_raw(only1, raw.data_ptr())
    root.synchronize()
    assert np.array_equal(root.unit_keys(), full.unit_keys())
    for k in np.concatenate([only0, only1]):
        sf, wf = full.read_unit(k)
        sr, wr = root.read_unit(k)
        assert np.array_equal(wf, wr) and np.array_equal(sf.view(np.uint32), sr.view(np.uint32)), "single-toucher unit %d is not bit-identical" % k
    for k in multi:
        sf, wf = full.read_unit(k)
        sr, wr = root.read_unit(k)
        assert np.array_equal(wf, wr) and float(np.abs(sf - sr).max()) <= 1e-5
    for v in parts + [full, merged]:
        v.close()


@pytest.mark.parametrize("impl, root", [("owner", -2), ("owner", 0), ("owner", -1), ("ring", 0), ("ring", -1)])
def test_frame_split_merge_with_four_loopback_ranks_on_one_gpu(gpu, impl, root):
    """The product's merge -- er_tsdf_allreduce: csrc/er_merge_protocol.h over the DEVICE volumes and their kernels -- with FOUR ranks on this one GPU
    (er_comm_create_loopback: host threads, device-to-device copies for the wire; RCCL refuses two ranks on one device).  Round 6: the OWNER merge
    (reduce-scatter by unit over band records, sums in rank order in the owner's kernel) with the result left distributed (-2), gathered on rank 0, or
    on every rank (-1); and round 5's ring protocol (ER_MERGE_IMPL=ring: whole planes through one sum, raw units point to point) for comparison.
    tests/helpers.py::check_frame_split_merge holds the assertions; tests/test_distributed_gpu.py runs the same checker over RCCL when the box has
    two or more GPUs."""
    from elasticreconstruction_amd import parallel
    out = helpers.check_frame_split_merge(lambda: parallel.LoopbackComms(4), [0, 0, 0, 0], root, impl, repeat=2 if (impl, root) == ("owner", -2) else 1)
    print("loopback merge:", out)
    if impl == "owner" and root == -2:
        assert out["moved_MB"] < 0.45 * out["ring_equivalent_MB"], out        # records, not planes


def test_owner_merge_failure_on_one_loopback_rank_reaches_all(gpu):
    """A failure only ONE rank sees must not leave the others waiting (the loopback steps now agree on their status inside their
    barriers; tests/cpp/merge_protocol_check.cpp injects the failures BEFORE a collective).  Here rank 1's volume is too small for the units the gather
    to every rank sends it: its unit pool runs out in the import after the last collective -- rank 1 reports it, every call returns."""
    from elasticreconstruction_amd import parallel
    blocks = helpers.merge_blocks(4, 50, [0, 0, 0, 0])
    vols = []
    for r, (sc, w) in enumerate(blocks):
        v = TSDFVolume(max_units=2048)
        v.IntegrateFrames(None, sc["traj"], w, device_ptr=sc["depth"].data_ptr())
        v.synchronize()
        vols.append(v)
    small = TSDFVolume(max_units=vols[1].unit_count() + 1)
    small.IntegrateFrames(None, blocks[1][0]["traj"], blocks[1][1], device_ptr=blocks[1][0]["depth"].data_ptr())
    small.synchronize()
    comms = parallel.LoopbackComms(4)
    res = comms.allreduce_failing([vols[0], small, vols[2], vols[3]], root=parallel.MERGE_ALL)
    assert res[1][0] != 0 and "pool" in res[1][1], res
    comms.close()
    for v in vols + [small]:
        v.close()


def test_config1_identity_trajectory_properties(gpu):
    """BASELINE.json config 1 shape at full size: 100 identical-pose frames into the 512-unit region.
    Size-independent properties: every weight is 0 or 100 (each frame updates the same voxel set),
    sum(weight) = 100 * N_upd(one frame, oracle), and a voxel fed the same tsdf 100 times keeps it
    to float rounding."""
    T = synth.look_at((1.5, 1.5, 1.5), (0.0, 0.0, 1.0))
    depth1 = synth.to_numpy_u16(synth.render_depth(T[None]))
    F = 100
    vol = TSDFVolume(max_units=600)
    vol.IntegrateFrames(np.repeat(depth1, F, axis=0), np.repeat(T[None], F, axis=0))
    ora = OracleVolume()
    ora.Integrate(depth1[0], T)
    assert np.array_equal(vol.unit_keys(), ora.unit_keys())
    assert vol.sum_weight() == F * ora.sum_weight()
    for k in vol.unit_keys()[::7]:
        s, w = vol.read_unit(k)
        so, wo = ora.read_unit(k)
        assert np.array_equal(w, wo * F)
        assert np.abs(s - so).max() <= 2e-5
    vol.close()


def test_non_vga_image_size(gpu):
    """Generic cols x rows (not multiples of the 32-pixel tiles, smaller than the reference's 640 x 480):
    rigid integration against the oracle, which takes the image size as a parameter."""
    cols, rows = 300, 210
    cam = np.array([250.0, 251.0, 149.5, 104.5, 2.5, 2.5], np.float32)
    poses = synth.circle_trajectory(3000)[9::650][:3]
    depth = synth.to_numpy_u16(synth.render_depth(poses, cols=cols, rows=rows, cam=tuple(cam[:4])))
    vol, ora = TSDFVolume(cols, rows, cam, max_units=256), OracleVolume(cols, rows, cam)
    assert np.array_equal(vol.ScaleDepth(depth[0]).view(np.uint32), ora.ScaleDepth(depth[0]).view(np.uint32))
    vol.IntegrateFrames(depth, poses)
    for i in range(3):
        ora.Integrate(depth[i], poses[i])
    assert helpers.assert_volumes_identical(vol, ora, "300x210") > 10
    vol.close()


def test_full_config2_batching_invariance(gpu):
    """BASELINE.json config 2 at FULL size (3000 frames, control-grid warp, 512^3 region): the result must not
    depend on how the stream is cut into batches -- 60 calls of 50 frames (bench.py's steps) vs one call that the
    library cuts into 64-frame launches vs a ragged cut -- and sum(weight) must equal the number of voxel updates
    reported per launch.  Bit-exact comparison of every unit."""
    import torch
    sc = synth.make_scenario(3000, interval=50, warp=True, device="cuda:0")
    warp = synth.warp_arrays(sc)
    depth = sc["depth"]
    torch.cuda.synchronize()
    px = depth.shape[1]

    def run(cuts):
        vol = TSDFVolume(max_units=640)
        for lo, hi in cuts:
            w = dict(ctr=warp["ctr"], resolution=warp["resolution"], length=warp["length"], grid_index=warp["grid_index"][lo:hi],
                     seg=warp["seg"][lo:hi], madj=warp["madj"][lo:hi])
            vol.IntegrateFrames(None, sc["traj"][lo:hi], w, device_ptr=depth.data_ptr() + lo * px * 2)
        return vol

    a = run([(s * 50, s * 50 + 50) for s in range(60)])
    b = run([(0, 3000)])
    edges = [0, 1, 7, 70, 71, 500, 1999, 3000]
    c = run(list(zip(edges[:-1], edges[1:])))
    n = helpers.assert_volumes_identical(a, b, "50-frame steps vs 64-frame launches")
    helpers.assert_volumes_identical(a, c, "50-frame steps vs ragged cuts")
    assert 100 <= n <= 512, "config 2 must stay inside the 512-unit (512^3) region, got %d units" % n
    assert a.sum_weight() == b.sum_weight() == c.sum_weight() > 4e9
    for v in (a, b, c):
        v.close()


@pytest.mark.gpu
def test_full_config2_equals_the_reference_build(gpu):
    """BASELINE.json configs[1] at FULL size against the REFERENCE's own code: tests/golden/config2_golden.json holds the digest of
    the volume that /root/reference/Integrate/*.cpp (compiled unmodified, driven through CIntegrateApp::Init / Execute on
    pose.log / seg.log / .ctr) leaves after all 3000 frames with the control-grid warp (tests/golden/make_golden_config2.py).
    The frames are re-rendered on the GPU from the committed camera poses (digest checked), integrated in bench.py's steps of
    150 frames, and unit keys, every sdf_ and every weight_ array must be bit-identical."""
    import hashlib
    import json
    import torch
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(gdir, "config2_inputs.npz"))
    g = json.load(open(os.path.join(gdir, "config2_golden.json")))
    n = int(g["frames"])
    depth = synth.render_depth(z["world"], device="cuda:0")
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for lo in range(0, n, 250):
        h.update(synth.to_numpy_u16(depth[lo:lo + 250]).tobytes())
    assert h.hexdigest() == g["depth_sha256"], "synthetic renderer is not bit-reproducible on this device"
    sc = dict(depth=depth, traj=z["traj"], pose=z["pose"], seg=z["seg"], grids=z["grids"], interval=int(z["meta"][0]),
              resolution=int(z["meta"][1]), length=float(z["length"]), n=n)
    warp = synth.warp_arrays(sc)
    px = depth.shape[1]
    vol = TSDFVolume(max_units=640)
    for lo in range(0, n, 150):
        hi = lo + 150
        w = dict(ctr=warp["ctr"], resolution=warp["resolution"], length=warp["length"], grid_index=warp["grid_index"][lo:hi],
                 seg=warp["seg"][lo:hi], madj=warp["madj"][lo:hi])
        vol.IntegrateFrames(None, sc["traj"][lo:hi], w, device_ptr=depth.data_ptr() + lo * px * 2)
    d = helpers.volume_digest(vol)
    assert d["keys"] == g["volume"]["keys"], "unit key sets differ: %d vs %d units" % (len(d["keys"]), len(g["volume"]["keys"]))
    assert d["sum_weight"] == g["volume"]["sum_weight"], (d["sum_weight"], g["volume"]["sum_weight"])
    assert d["sha256"] == g["volume"]["sha256"], "volume differs from the reference build's"
    vol.close()


class _DictVolume:
    """{key: (sdf_, weight_)} behind the unit_keys / read_unit surface helpers.volume_digest and assert_volumes_identical use."""

    def __init__(self, units):
        self.u = units

    def unit_keys(self):
        return np.array(sorted(self.u), np.int32)

    def read_unit(self, k):
        s, w = self.u[int(k)]
        return np.ascontiguousarray(s, np.float32), np.ascontiguousarray(w, np.float32)


def _sampled_job_against_the_reference(tmp_path, tag, n_frames, n_runs, run_len, max_units, **scene):
    """A sampled stream of a LONG Integrate job -- n_runs runs of run_len consecutive frames spread from the first to the last
    fragment of an n_frames job, every frame with its TRUE frame id, i.e. its own lattice of the job's full .ctr and its own entry of
    the full trajectory -- through the host mirror of CIntegrateApp (-> er_tsdf_integrate_frames with the warp) and through the
    REFERENCE's own CIntegrateApp::Execute (oracle/_ref/libref_tsdf.so, Integrate/*.cpp compiled in place), both reading the same
    pose.log / seg.log / g.ctr.  Unit key sets equal, every sdf_ / weight_ bit pattern equal; where the reference build is absent the
    committed digest of tests/golden/configs34_golden.json (tests/golden/make_golden_configs34.py, same files + depth digests) stands in."""
    import hashlib
    import json
    from oracle import pyoracle, refcheck
    ids = refcheck.sampled_frames(n_frames, 50, n_runs, run_len)
    sc = synth.make_scenario(n_frames, interval=50, warp=True, revolutions=n_frames / 3000.0, render_frames=ids, device="cuda:0", **scene)
    depth = synth.to_numpy_u16(sc["depth"])
    paths = refcheck.write_integrate_files(sc, str(tmp_path))
    app = IntegrateApp(max_units=max_units)
    app.pose_filename_, app.seg_filename_, app.ctr_filename_ = paths
    app.ctr_num_, app.ctr_resolution_, app.ctr_length_, app.ctr_interval_ = n_frames // 50, sc["resolution"], sc["length"], 50
    app.Init()
    for k, f in enumerate(ids):
        app.Execute(int(f) + 1, depth[k])
    app.Finish(save=False)
    assert not app.exit_ and app.frames_integrated == len(ids)
    flags, skipped = app.volume_.status()
    assert flags == 0 and skipped == 0, "overflow flags %d, %d pixels beyond +-96 m" % (flags, skipped)
    checked = []
    if pyoracle.have_ref():
        ref_units, _, _ = refcheck.reference_volume_of_frames(sc, depth, ids, str(tmp_path))
        n = helpers.assert_volumes_identical(app.volume_, _DictVolume(ref_units), tag + " vs the reference build")
        checked.append("reference")
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "configs34_golden.json")))[tag]
    assert g["frame_ids_sha256"] == hashlib.sha256(ids.tobytes()).hexdigest()
    inputs = {os.path.basename(p): hashlib.sha256(open(p, "rb").read()).hexdigest() for p in paths}
    inputs["depth"] = hashlib.sha256(depth.tobytes()).hexdigest()
    if inputs == g["inputs"]:                         # same text files and images as the golden run saw: its digest must come out
        d = helpers.volume_digest(app.volume_)
        assert d["keys"] == g["volume"]["keys"] and d["sum_weight"] == g["volume"]["sum_weight"] and d["sha256"] == g["volume"]["sha256"], \
            tag + ": volume differs from the golden digest of the reference build"
        checked.append("golden")
    assert checked, "neither oracle/_ref nor bit-reproducible golden inputs on this host: %s" % sorted(k for k in inputs if inputs[k] != g["inputs"][k])
    keys = app.volume_.unit_keys()
    app.volume_.close()
    return keys, checked


def test_config4_hashed_grid_stream_equals_the_reference_build(gpu, tmp_path):
    """BASELINE.json configs[3]'s scene (bench.py --config 4: 10 000 frames on a drifting path through a 6 m room, 200 control
    lattices): 400 frames sampled from the first to the last fragment.  This is the part of the reference the 512^3 scenes never
    reach: unit index arithmetic at NEGATIVE world coordinates (the +256*64 offset, TSDFVolume.cpp:50-53), keys of hash_key
    (TSDFVolume.h:62-64) outside the 8x8x8 region, a hashed grid that grows past 512 units, and warped pixels the fragment's lattice
    rejects (ControlGrid.h:51-54: most of the room is outside the fragment's 3 m cube)."""
    from oracle import refcheck
    keys, checked = _sampled_job_against_the_reference(tmp_path, "config4", 10000, 40, 10, 4096, radius_drift=1.5, room=(-1.5, 4.5))
    c = refcheck.unit_coordinates(keys)
    assert len(keys) > 512, "the scene must grow the hashed grid past 512 units, got %d" % len(keys)
    assert int((c < 0).any(axis=1).sum()) > 100 and c.min() <= -4 and c.max() >= 12, "negative / far unit coordinates were not reached"
    print("config4 sampled stream: %d units, coordinates %s .. %s, checked against %s" % (len(keys), c.min(0), c.max(0), checked))


def test_config5_100_lattice_stream_equals_the_reference_build(gpu, tmp_path):
    """BASELINE.json configs[4]'s integrate half (bench.py --config 5: 5000 frames, 100 control lattices, 1.67 revolutions): 200
    frames sampled from the first to the last fragment, each warped with ITS lattice of the 100 (IntegrateApp.cpp:241)."""
    keys, checked = _sampled_job_against_the_reference(tmp_path, "config5", 5000, 20, 10, 1024)
    assert 100 <= len(keys) <= 512
    print("config5 sampled stream: %d units, checked against %s" % (len(keys), checked))


@pytest.mark.gpu
def test_exact_arithmetic_cores_exhaustive(gpu):
    """The voxel update replaces hipcc's IEEE '/' and sqrtf by their un-wrapped cores and the float64 pixel rounding
    by a float32 form, and the two divisions by constants (/ tsdf_trunc_, / 1000.f) by a multiply + Markstein correction
    (er_tsdf_math.h: div2_inrange, div_inrange, sqrt_inrange, pixel_index, band_quotient_core, div1000_core).  tests/hip/arith_check
    compares them ON THE GPU with the plain operators: every float for sqrt and pixel rounding (two image limits),
    2^31 hashed operand triples per division scenario.  Zero mismatches required."""
    import subprocess
    import __graft_entry__ as g
    exe = g._build_arith_check()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l.split() for l in r.stdout.strip().splitlines()]
    assert len(lines) == 7 and all(int(l[2]) > 10 ** 9 and int(l[4]) == 0 for l in lines), r.stdout


def test_randomised_configurations_bit_exact(gpu):
    """Seeded fuzz over what the kernels' exactness arguments depend on: image size (not tile multiples), intrinsics
    (principal point off-centre / zero, anisotropic focal lengths), camera poses anywhere around and inside the room
    (voxels behind the camera, grazing views, the camera inside a unit), depth images with holes, salt noise and values
    beyond integration_trunc, and a warp through randomly deformed control grids.  Every configuration must give the
    oracle's unit set and bit patterns."""
    rng = np.random.default_rng(int(os.environ.get("ER_FUZZ_SEED", "20240919")))      # (ER_FUZZ_SEED / ER_FUZZ_CASES: one-off wider sweeps, profiles/r06x_*)
    for case in range(int(os.environ.get("ER_FUZZ_CASES", "10"))):
        warped = case % 2 == 1                                # odd cases go through Reproject, whose XYZ2UVD bounds are the
        if warped:                                            # literal 640 x 480 (TSDFVolume.h:55): full-size images only
            cols, rows = 640, 480
            fx, fy = float(rng.uniform(400, 650)), float(rng.uniform(400, 650))
        else:
            cols, rows = int(rng.integers(48, 200)), int(rng.integers(40, 160))
            fx, fy = float(rng.uniform(60, 260)), float(rng.uniform(60, 260))
        cx = 0.0 if case == 3 else float(rng.uniform(0.2, 0.8) * cols)
        cy = float(rng.uniform(0.2, 0.8) * rows)
        cam = np.array([fx, fy, cx, cy, 2.5, float(rng.uniform(1.0, 3.5))], np.float32)
        n = 3 if warped else int(rng.integers(3, 9))
        poses = []
        for _ in range(n):
            eye = rng.uniform(0.3, 2.7, 3)
            tgt = rng.uniform(0.0, 3.0, 3)
            P = synth.look_at(tuple(eye), tuple(tgt))
            poses.append(P @ synth.perturbation(int(rng.integers(1 << 30)), 20.0, 0.0))     # roll / off-axis views
        poses = np.stack(poses)
        depth = synth.to_numpy_u16(synth.render_depth(poses, cols=cols, rows=rows, cam=tuple(float(c) for c in cam[:4]))).copy()
        holes = rng.random(depth.shape) < 0.05
        depth[holes] = 0
        salt = rng.random(depth.shape) < 0.01
        depth[salt] = rng.integers(1, 12000, int(salt.sum()), dtype=np.uint16)       # up to 12 m: beyond integration_trunc, units still touched
        vol, ora = TSDFVolume(cols, rows, cam, max_units=4096), OracleVolume(cols, rows, cam)
        warp = None
        if warped:
            res, length = 8, 3.0
            k, j, i = np.meshgrid(np.arange(res + 1), np.arange(res + 1), np.arange(res + 1), indexing="ij")
            base = np.stack([i.ravel(), j.ravel(), k.ravel()], 1) * (length / res)
            grid = (base + rng.normal(0, 0.01, base.shape)).astype(np.float32)
            seg = np.stack([np.linalg.inv(poses[0]) @ P for P in poses])      # camera poses in the first frame's fragment cube
            cube = synth.basepose()
            seg = np.stack([cube @ S for S in seg])
            madj = np.stack([np.linalg.inv(poses[f]) @ poses[0] @ np.linalg.inv(seg[0]) for f in range(n)])
            warp = dict(ctr=grid[None], resolution=res, length=np.float32(length), grid_index=np.zeros(n, np.int32), seg=seg, madj=madj)
        vol.IntegrateFrames(depth, poses, warp)
        for f in range(n):
            d = depth[f]
            if warp is not None:
                d = ora.Reproject(d, warp["ctr"][0], warp["resolution"], warp["length"], warp["seg"][f], warp["madj"][f])
            ora.Integrate(d, poses[f])
        helpers.assert_volumes_identical(vol, ora, "fuzz case %d (%dx%d)" % (case, cols, rows))
        vol.close()


def test_reset_unit_shard_pinned_input_and_status(gpu):
    """The round-2 additions of the ABI on the golden warp scene:
       * er_tsdf_reset          -> an emptied volume integrates to the same golden digest again;
       * er_tsdf_set_unit_shard -> 3 volumes that each own a third of the units (bit-exact alternative):
                                   disjoint key sets, union == single-GPU key set, every unit bit-identical;
       * host frames in page-locked memory (own copy stream, double-buffered staging) -> same digest;
       * er_tsdf_status         -> reports an exhausted unit pool without waiting for a read-back."""
    sc = helpers.golden_warp()
    g = helpers.golden()["warp"]
    depth = synth.to_numpy_u16(sc["depth"])
    warp = synth.warp_arrays(sc)
    vol = TSDFVolume(max_units=256)
    vol.IntegrateFrames(depth, sc["traj"], warp)
    assert helpers.volume_digest(vol)["sha256"] == g["sha256"]
    assert vol.status() == (0, 0)
    vol.reset()
    assert vol.unit_count() == 0 and vol.sum_weight() == 0.0
    # second life of the same handle, this time from page-locked host memory, several frames per call
    arena = _ffi.PinnedArena()
    arena.reset(depth.nbytes + 8192)
    pinned = arena.take(depth.shape, np.uint16)
    pinned[...] = depth
    vol.IntegrateFrames(pinned, sc["traj"], warp)
    d = helpers.volume_digest(vol)
    assert d["keys"] == g["keys"] and d["sha256"] == g["sha256"]
    # unit shard
    world = 3
    shards = [TSDFVolume(max_units=256) for _ in range(world)]
    seen = {}
    for r, sv in enumerate(shards):
        sv.set_unit_shard(r, world)
        sv.IntegrateFrames(depth, sc["traj"], warp)
        for k in sv.unit_keys():
            assert int(k) not in seen and _ffi.lib().er_unit_owner(int(k), world) == r
            seen[int(k)] = sv
    assert sorted(seen) == g["keys"]
    for k, sv in seen.items():
        s0, w0 = vol.read_unit(k)
        s1, w1 = sv.read_unit(k)
        assert np.array_equal(w0, w1) and np.array_equal(s0.view(np.uint32), s1.view(np.uint32)), "unit %d differs in shard mode" % k
    with pytest.raises(_ffi.ErError, match="already holds"):
        shards[0].set_unit_shard(1, 2)
    for sv in shards:
        sv.close()
    vol.close()
    arena.close()
    small = TSDFVolume(max_units=4)
    small.IntegrateFrames(depth[:2], sc["traj"][:2])
    small.synchronize()
    flags, _ = small.status()
    assert flags & 1, "exhausted pool not reported by er_tsdf_status"
    with pytest.raises(_ffi.ErError, match="pool exhausted"):
        small.unit_count()
    small.close()


def _surface_oracle(units):
    """CPU restatement of er_tsdf_extract_surface over {key: (sdf, weight)} (numpy float32, same operation order): for every
    observed voxel and its +x/+y/+z neighbour (also across unit borders) with strictly opposite sdf signs, the crossing point
    p = position + F / (F - Fn) * voxel size; units ascending, voxels i,j,k, axes x,y,z."""
    ul = 3.0 / 512.0
    ulf = np.float32(ul)
    out = []
    for key in sorted(units):
        xi, yi, zi = key >> 18, (key >> 9) & 511, key & 511
        F = units[key][0].reshape(64, 64, 64)
        W = units[key][1].reshape(64, 64, 64)

        def nb(dk, axis):
            Fn, Wn = np.zeros_like(F), np.zeros_like(W)
            sl_to = [slice(None)] * 3
            sl_from = [slice(None)] * 3
            sl_to[axis], sl_from[axis] = slice(0, 63), slice(1, 64)
            Fn[tuple(sl_to)], Wn[tuple(sl_to)] = F[tuple(sl_from)], W[tuple(sl_from)]
            if dk in units:
                F2, W2 = units[dk][0].reshape(64, 64, 64), units[dk][1].reshape(64, 64, 64)
                sl_to[axis], sl_from[axis] = 63, 0
                Fn[tuple(sl_to)], Wn[tuple(sl_to)] = F2[tuple(sl_from)], W2[tuple(sl_from)]
            return Fn, Wn
        nbs = [nb(key + 512 * 512 if xi < 511 else -1, 0), nb(key + 512 if yi < 511 else -1, 1), nb(key + 1 if zi < 511 else -1, 2)]
        cross = np.stack([(W != 0) & (Wn != 0) & (((F > 0) & (Fn < 0)) | ((F < 0) & (Fn > 0))) for Fn, Wn in nbs], axis=3)
        ii, jj, kk, aa = np.nonzero(cross)
        if ii.size == 0:
            continue
        g = np.stack([(ii + (xi - 256) * 64), (jj + (yi - 256) * 64), (kk + (zi - 256) * 64)], 1)
        p = (g.astype(np.float64) * ul).astype(np.float32)
        Fv = F[ii, jj, kk]
        Fnv = np.choose(aa, [nbs[0][0][ii, jj, kk], nbs[1][0][ii, jj, kk], nbs[2][0][ii, jj, kk]])
        t = (Fv / (Fv - Fnv)).astype(np.float32) * ulf
        p[np.arange(ii.size), aa] = p[np.arange(ii.size), aa] + t
        out.append(np.concatenate([p, aa[:, None].astype(np.float32)], 1))
    return np.concatenate(out) if out else np.zeros((0, 4), np.float32)


def test_zero_crossing_extraction_matches_cpu_restatement(gpu):
    """er_tsdf_extract_surface: the point list equals a numpy restatement element for element (same float32
    operations, same order, neighbours across unit borders included), and the points lie on the scene's surfaces within a
    voxel.  Scene: the golden poses in a room whose far walls sit at 2.62207 m = exactly between voxel 63 of unit 6 and voxel
    0 of unit 7 (447.5 voxels), so that every wall crossing is a CROSS-UNIT crossing (the millimetre quantisation of the depth
    scatters the wall's points over both units, so both get allocated)."""
    poses, _ = helpers.golden_rigid()
    wall = 447.5 * 3.0 / 512.0
    depth = synth.to_numpy_u16(synth.render_depth(poses, hi=wall, sphere=False))
    vol = TSDFVolume(max_units=256)
    vol.IntegrateFrames(depth, poses)
    got = vol.extract_surface()
    units = {int(k): vol.read_unit(k) for k in vol.unit_keys()}
    want = _surface_oracle(units)
    assert got.shape == want.shape and got.shape[0] > 20000, (got.shape, want.shape)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "%d rows differ" % int((got != want).any(1).sum())
    # most crossings come from unit borders here: local index 63 along the crossing's own axis, against the neighbouring unit
    ul = 3.0 / 512.0
    idx = np.floor(got[:, :3] / ul + 1e-3).astype(np.int64) % 64
    across = ((idx == 63) & (got[:, 3:4] == np.arange(3)[None, :])).any(1)
    assert across.sum() > 10000, "only %d of %d crossings are cross-unit ones" % (across.sum(), got.shape[0])
    # geometry: the crossings lie on the walls (within two voxels)
    dw = np.minimum(np.abs(got[:, :3] - synth.ROOM_LO), np.abs(got[:, :3] - wall)).min(1)
    assert (dw < 2.0 * ul).mean() > 0.9, "only %.1f %% of the zero crossings lie on a surface" % (100 * (dw < 2.0 * ul).mean())
    vol.close()


def _mesh_oracle(units):
    """numpy restatement of k_mesh (er_tsdf_extract_mesh): the same case table (er_mc_table; its properties are checked on
    the CPU by tests/test_mc_table.py), cells valid iff all eight voxels are observed, inside iff sdf < 0, vertices
    pos(L) + (F_L / (F_L - F_H)) * voxel size from the edge's lower voxel in float32; units by ascending key, cells in
    i, j, k order, triangles in table order.  units: {key: (sdf[262144], weight[262144])}."""
    import ctypes as C
    from elasticreconstruction_amd import _ffi
    tab = np.zeros(256 * 16, np.uint8)
    assert _ffi.lib().er_mc_table(tab.ctypes.data_as(C.c_void_p)) == 0
    tab = tab.reshape(256, 16)
    ul = 3.0 / 512.0
    ulf = np.float32(ul)
    out = []
    for key in sorted(units):
        xi, yi, zi = key >> 18, (key >> 9) & 511, key & 511
        S = np.zeros((65, 65, 65), np.float32)
        W = np.zeros((65, 65, 65), np.float32)
        for dx in range(2):
            for dy in range(2):
                for dz in range(2):
                    k2 = key + dx * 512 * 512 + dy * 512 + dz
                    if k2 not in units or xi + dx > 511 or yi + dy > 511 or zi + dz > 511:
                        continue
                    s, w = (a.reshape(64, 64, 64) for a in units[k2])
                    sl = tuple(slice(64, 65) if d else slice(0, 64) for d in (dx, dy, dz))
                    src = tuple(slice(0, 1) if d else slice(0, 64) for d in (dx, dy, dz))
                    S[sl], W[sl] = s[src], w[src]
        corner = lambda A, c: A[(c & 1):(c & 1) + 64, ((c >> 1) & 1):((c >> 1) & 1) + 64, (c >> 2):(c >> 2) + 64]
        valid = np.ones((64, 64, 64), bool)
        case = np.zeros((64, 64, 64), np.int32)
        for c in range(8):
            valid &= corner(W, c) != 0
            case |= (corner(S, c) < 0).astype(np.int32) << c
        ntri = (tab[case][..., 0::3][..., :5] != 255).sum(-1) * valid
        cells = np.argwhere(ntri > 0)                                        # C order = i, j, k order
        if not len(cells):
            continue
        ci, cj, ck = cells.T
        cs = case[ci, cj, ck]
        nt = ntri[ci, cj, ck]
        verts = np.zeros((len(cells), 15, 3), np.float32)
        G = [((np.arange(66) + (q - 256) * 64).astype(np.float64) * ul).astype(np.float32) for q in (xi, yi, zi)]
        for t in range(15):
            e = tab[cs, t].astype(np.int32)
            live = 3 * nt > t
            e = np.where(live, e, 0)
            axis, u, v = e >> 2, e & 1, (e >> 1) & 1
            a0 = np.where(axis == 0, 0, u)
            b0 = np.where(axis == 1, 0, np.where(axis == 0, u, v))
            c0 = np.where(axis == 2, 0, v)
            li, lj, lk = ci + a0, cj + b0, ck + c0
            hi_, hj, hk = li + (axis == 0), lj + (axis == 1), lk + (axis == 2)
            fl, fh = S[li, lj, lk], S[hi_, hj, hk]
            with np.errstate(divide="ignore", invalid="ignore"):
                tt = (fl / (fl - fh)).astype(np.float32)
            p = np.stack([G[0][li], G[1][lj], G[2][lk]], 1)
            add = (tt * ulf).astype(np.float32)
            for q in range(3):
                p[:, q] = np.where(axis == q, (p[:, q] + add).astype(np.float32), p[:, q])
            verts[:, t] = np.where(live[:, None], p, 0)
        keep = np.arange(15)[None, :] < (3 * nt)[:, None]
        out.append(verts[keep].reshape(-1, 3, 3))
    return np.concatenate(out) if out else np.zeros((0, 3, 3), np.float32)


def test_marching_cubes_mesh_matches_cpu_restatement_and_is_watertight(gpu):
    """er_tsdf_extract_mesh (triangle connectivity): the triangle soup equals the numpy restatement bit for bit
    (same case table, same float32 vertex expression, same order, cells across unit borders included); welded by exact vertex
    equality every mesh edge is shared by exactly two triangles with opposite directions except on the border of the observed
    region; every vertex with two non-zero end values is one of er_tsdf_extract_surface's zero crossings; and the triangles lie
    on the scene's surfaces.  Scene as in the zero-crossing test: walls exactly between two units."""
    poses, _ = helpers.golden_rigid()
    wall = 447.5 * 3.0 / 512.0
    depth = synth.to_numpy_u16(synth.render_depth(poses, hi=wall, sphere=True))
    vol = TSDFVolume(max_units=256)
    vol.IntegrateFrames(depth, poses)
    got = vol.extract_mesh()
    units = {int(k): vol.read_unit(k) for k in vol.unit_keys()}
    want = _mesh_oracle(units)
    assert got.shape == want.shape and got.shape[0] > 40000, (got.shape, want.shape)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "%d triangles differ" % int((got != want).any((1, 2)).sum())
    # weld by exact vertex equality
    flat = got.reshape(-1, 3)
    uniq, inv = np.unique(flat.view(np.uint32).reshape(-1, 3), axis=0, return_inverse=True)
    tri = inv.reshape(-1, 3)
    e = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]])
    e = e[e[:, 0] != e[:, 1]]                                                # (a vertex exactly on a voxel collapses a side)
    und = np.sort(e, 1)
    code = und[:, 0].astype(np.int64) * len(uniq) + und[:, 1]
    vals, cnt = np.unique(code, return_counts=True)
    assert (cnt <= 2).all(), "an edge is shared by %d triangles" % cnt.max()
    assert (cnt == 2).mean() > 0.97, "only %.1f %% of the edges are interior" % (100 * (cnt == 2).mean())
    # orientation: the two triangles of an interior edge traverse it in opposite directions
    dcode = e[:, 0].astype(np.int64) * len(uniq) + e[:, 1]
    assert len(np.unique(dcode)) == len(dcode), "a directed edge appears twice: inconsistent winding"
    # vertices are zero crossings of er_tsdf_extract_surface (those with strictly opposite end signs)
    surf = vol.extract_surface()[:, :3]
    sset = set(map(bytes, np.ascontiguousarray(surf).view(np.uint8).reshape(len(surf), 12)))
    vset = list(map(bytes, np.ascontiguousarray(uniq.view(np.float32)).view(np.uint8).reshape(len(uniq), 12)))
    hit = sum(1 for b in vset if b in sset)
    assert hit > 0.95 * len(vset), "%d of %d mesh vertices are zero crossings" % (hit, len(vset))
    # geometry: on the walls (or the sphere, when a pose sees it) within two voxels; normals face the free space = the room's interior
    ul = 3.0 / 512.0
    cen = got.mean(1)
    dlo, dhi = np.abs(cen - synth.ROOM_LO), np.abs(cen - wall)
    dw = np.minimum(dlo, dhi).min(1)
    ds = np.abs(np.linalg.norm(cen - np.array(synth.SPHERE_C), axis=1) - synth.SPHERE_R)
    assert (np.minimum(dw, ds) < 2.0 * ul).mean() > 0.9
    nrm = np.cross(got[:, 1] - got[:, 0], got[:, 2] - got[:, 0])
    flat_wall = (dw < 0.5 * ul) & (ds > 0.2)                                   # clearly on ONE wall, away from the sphere and the corners
    ax = np.minimum(dlo, dhi)[flat_wall].argmin(1)
    second = np.sort(np.minimum(dlo, dhi)[flat_wall], 1)[:, 1]
    inward = np.where(dlo[flat_wall, ax] < dhi[flat_wall, ax], 1.0, -1.0)      # at the low wall the interior lies towards +axis
    ok = (nrm[flat_wall, ax] * inward > 0)[second > 0.1]
    assert ok.size > 10000 and ok.mean() > 0.999, "%d wall triangles, %.2f %% face the interior" % (ok.size, 100 * ok.mean())
    vol.close()


def test_marching_cubes_of_an_analytic_sphere_is_a_closed_genus_0_surface(gpu):
    """An implementation-independent check of er_tsdf_extract_mesh (missing 7: the numpy restatement and the kernel share one
    author and one table): a volume that holds the truncated signed distance of an ANALYTIC sphere -- written straight into 2 x 2 x 2 units
    through er_tsdf_import_weighted, every voxel observed, the surface crossing all three unit borders off the lattice's symmetry planes --
    must come out as a closed orientable surface of genus 0: welded by exact vertex equality, every edge is shared by exactly two triangles
    that traverse it in opposite directions, V - E + F = 2; every vertex lies on the sphere to a fraction of a voxel (linear interpolation of
    a distance field: second-order error), the area is 4 pi r^2 to 0.2 %, the enclosed volume 4/3 pi r^3 to 0.2 %, and every normal points
    away from the centre (towards positive distance = free space)."""
    import torch
    ul = 3.0 / 512.0
    c = np.array([0.371, 0.383, 0.377])
    r = 0.25
    keys, planes = [], []
    ax = np.arange(64, dtype=np.float64)
    for xi in (256, 257):
        for yi in (256, 257):
            for zi in (256, 257):
                keys.append(xi * 512 * 512 + yi * 512 + zi)
                gx = (np.float32((xi - 256) * 64 * ul) + ax * ul).astype(np.float32)       # I2F + grid coordinate, TSDFVolume.h:66-68 / .cpp:75
                gy = (np.float32((yi - 256) * 64 * ul) + ax * ul).astype(np.float32)
                gz = (np.float32((zi - 256) * 64 * ul) + ax * ul).astype(np.float32)
                d = np.sqrt((gx[:, None, None] - c[0]) ** 2 + (gy[None, :, None] - c[1]) ** 2 + (gz[None, None, :] - c[2]) ** 2) - r
                sdf = np.clip(d / 0.03, -1.0, 1.0).astype(np.float32).reshape(-1)           # l = (i * 64 + j) * 64 + k
                planes.append(np.stack([sdf, np.ones_like(sdf)]))                           # sdf * weight | weight, weight = 1: every voxel observed
    order = np.argsort(keys)
    keys = np.array(keys, np.int32)[order]
    buf = torch.from_numpy(np.stack(planes)[order]).to("cuda:0")
    vol = TSDFVolume(max_units=16)
    vol.import_weighted(keys, buf.data_ptr())
    vol.synchronize()
    tri = vol.extract_mesh()
    assert tri.shape[0] > 20000
    flat = tri.reshape(-1, 3)
    uniq, inv = np.unique(flat.view(np.uint32).reshape(-1, 3), axis=0, return_inverse=True)
    t = inv.reshape(-1, 3)
    keep = (t[:, 0] != t[:, 1]) & (t[:, 1] != t[:, 2]) & (t[:, 0] != t[:, 2])               # (a vertex exactly on a voxel collapses a triangle to a sliver)
    assert keep.mean() > 0.999
    t = t[keep]
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    nv = len(uniq)
    und = np.sort(e, 1)
    _, cnt = np.unique(und[:, 0].astype(np.int64) * nv + und[:, 1], return_counts=True)
    assert (cnt == 2).all(), "%d edges are not shared by exactly two triangles (closed surface)" % int((cnt != 2).sum())
    assert len(np.unique(e[:, 0].astype(np.int64) * nv + e[:, 1])) == len(e), "a directed edge appears twice: inconsistent winding"
    used = np.unique(t)
    assert len(used) - len(cnt) + len(t) == 2, "Euler characteristic %d (a sphere has 2)" % (len(used) - len(cnt) + len(t))
    P = uniq.view(np.float32).astype(np.float64)
    dev = np.abs(np.linalg.norm(P[used] - c, axis=1) - r)
    assert dev.max() < 0.05 * ul, "a vertex is %.3g voxels off the sphere" % (dev.max() / ul)
    A, B, C = P[t[:, 0]], P[t[:, 1]], P[t[:, 2]]
    n = np.cross(B - A, C - A)
    area = 0.5 * np.linalg.norm(n, axis=1).sum()
    volume = np.einsum("ij,ij->i", A - c, n).sum() / 6.0
    assert abs(area / (4 * np.pi * r * r) - 1) < 2e-3 and abs(abs(volume) / (4.0 / 3.0 * np.pi * r ** 3) - 1) < 2e-3, (area, volume)
    outward = np.einsum("ij,ij->i", n, (A + B + C) / 3.0 - c) > 0
    assert outward.all(), "%d triangles face the centre" % int((~outward).sum())
    vol.close()
