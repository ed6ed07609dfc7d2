"""CPU check of the arithmetic the HIP kernels execute: elasticreconstruction_amd/csrc/er_tsdf_math.h is
compiled for the host into a test-only library (tests/hostcheck) and compared bit for bit with the
oracle on the golden inputs.  This validates the shared per-voxel / per-pixel expressions (including the
exact shortcuts: truncation-band-only float64 division, float64 range tests before int conversion) here,
where there is no GPU; the -m gpu tests then validate the real kernels."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers
from elasticreconstruction_amd import synth, tsdf
from oracle.pyoracle import OracleVolume

HERE = os.path.dirname(os.path.abspath(__file__))
vp = C.c_void_p


@pytest.fixture(scope="module")
def hc():
    src = os.path.join(HERE, "hostcheck", "tsdf_hostcheck.cpp")
    out = os.path.join(HERE, "hostcheck", "_build", "libtsdf_hostcheck.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hdr = os.path.join(HERE, "..", "elasticreconstruction_amd", "csrc", "er_tsdf_math.h")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", src, "-o", out], check=True)
    L = C.CDLL(out)
    L.hc_create.restype = vp
    L.hc_create.argtypes = [C.c_int, C.c_int, vp]
    L.hc_destroy.argtypes = [vp]
    L.hc_scale_depth.argtypes = [vp, vp, vp]
    L.hc_reproject.argtypes = [vp, vp, vp, C.c_int, C.c_float, vp, vp]
    L.hc_integrate_frames.argtypes = [vp, C.c_int, vp, vp, vp]
    L.hc_unit_count.argtypes = [vp]
    L.hc_culled.restype = C.c_long
    L.hc_culled.argtypes = [vp]
    L.hc_kept.restype = C.c_long
    L.hc_kept.argtypes = [vp]
    L.hc_inside.restype = C.c_long
    L.hc_inside.argtypes = [vp]
    L.hc_inside_violations.restype = C.c_long
    L.hc_inside_violations.argtypes = [vp]
    for name in ("hc_sure", "hc_visited", "hc_unsure_pf", "hc_sure_violations", "hc_full", "hc_full_violations", "hc_exact_rows", "hc_exact_rows_needing"):
        getattr(L, name).restype = C.c_long
        getattr(L, name).argtypes = [vp]
    L.hc_set_patch_shape.restype = None
    L.hc_set_patch_shape.argtypes = [C.c_int]
    L.hc_touch_key_stress.restype = C.c_long
    L.hc_touch_key_stress.argtypes = [C.c_ulonglong, C.c_long, C.POINTER(C.c_long)]
    L.hc_full_stress.restype = C.c_long
    L.hc_full_stress.argtypes = [C.c_ulonglong, C.c_long, C.POINTER(C.c_long)]
    L.hc_sure_stress.restype = C.c_long
    L.hc_sure_stress.argtypes = [C.c_ulonglong, C.c_long, C.POINTER(C.c_long)]
    L.hc_unit_keys.argtypes = [vp, vp]
    L.hc_read_unit.argtypes = [vp, C.c_int, vp, vp]
    return L


class HcVolume:
    def __init__(self, L, cam=None, cols=640, rows=480):
        self.L = L
        self.cam = np.array([525.0, 525.0, 319.5, 239.5, 2.5, 2.5], np.float32) if cam is None else np.asarray(cam, np.float32)
        self.h = vp(L.hc_create(cols, rows, self.cam.ctypes.data_as(vp)))

    def integrate(self, depth, T):
        depth = np.ascontiguousarray(depth, np.uint16)
        T = np.ascontiguousarray(T, np.float64).reshape(-1, 16)
        Ti = np.ascontiguousarray(np.stack([tsdf._inverse(t) for t in T.reshape(-1, 4, 4)]).reshape(-1, 16))
        assert self.L.hc_integrate_frames(self.h, T.shape[0], depth.ctypes.data_as(vp), T.ctypes.data_as(vp), Ti.ctypes.data_as(vp)) == 0

    def reproject(self, depth, ctr, res, length, seg, madj):
        d = np.array(depth, np.uint16)
        g = np.ascontiguousarray(ctr, np.float32)
        s, m = np.ascontiguousarray(seg, np.float64), np.ascontiguousarray(madj, np.float64)
        self.L.hc_reproject(self.h, d.ctypes.data_as(vp), g.ctypes.data_as(vp), res, C.c_float(length), s.ctypes.data_as(vp), m.ctypes.data_as(vp))
        return d

    def scale(self, depth):
        d = np.ascontiguousarray(depth, np.uint16)
        out = np.empty(d.size, np.float32)
        self.L.hc_scale_depth(self.h, d.ctypes.data_as(vp), out.ctypes.data_as(vp))
        return out

    def unit_keys(self):
        n = self.L.hc_unit_count(self.h)
        k = np.empty(n, np.int32)
        self.L.hc_unit_keys(self.h, k.ctypes.data_as(vp))
        return k

    def read_unit(self, key):
        s, w = np.empty(64 ** 3, np.float32), np.empty(64 ** 3, np.float32)
        assert self.L.hc_read_unit(self.h, int(key), s.ctypes.data_as(vp), w.ctypes.data_as(vp)) == 0
        return s, w


def test_device_math_rigid_matches_golden(hc):
    poses, depth = helpers.golden_rigid()
    g = helpers.golden()
    v = HcVolume(hc)
    assert helpers.digest(v.scale(depth[0])) == g["scale_depth_frame0"]
    v.integrate(depth[:4], poses[:4])        # two "batches": masks / ordering across calls
    v.integrate(depth[4:], poses[4:])
    d = helpers.volume_digest(v)
    assert d["keys"] == g["rigid"]["keys"] and d["sha256"] == g["rigid"]["sha256"]
    # the exact (patch, frame) culling must both fire and change nothing
    culled, kept = hc.hc_culled(v.h), hc.hc_kept(v.h)
    assert culled > 0.15 * (culled + kept), "patch culling removed only %d of %d patch-frames" % (culled, culled + kept)
    # ... and so must the "whole patch inside the image" verdict that lets k_integrate skip the per-voxel range tests:
    # every voxel it covers was re-tested with the full voxel_project (same pixel, in range) by the host build
    inside = hc.hc_inside(v.h)
    print("patch-frames: %d culled, %d kept, %d of them inside (%.0f %%)" % (culled, kept, inside, 100.0 * inside / kept))
    assert hc.hc_inside_violations(v.h) == 0 and inside > 0.4 * kept
    assert hc.hc_full_violations(v.h) == 0
    print("full (patch, frame) visits: %d of %d kept (%.1f %%)" % (hc.hc_full(v.h), kept, 100.0 * hc.hc_full(v.h) / max(kept, 1)))


@pytest.mark.parametrize("shape", [0, 1, 2])
def test_device_math_warp_matches_golden(hc, shape):
    """shape 1: the 4 x 8 x 8 box k_integrate gives a wave (patch_may_update_box with eight corners); shape 2: the 8 x 8 x 8 cube the
    kernel used for part of round 3; shape 0: a 16 x 16 square of one slab.  Same golden digests every way: the culling and the
    shortcuts are exact whatever the patch."""
    hc.hc_set_patch_shape(shape)
    try:
        _warp_matches_golden(hc)
    finally:
        hc.hc_set_patch_shape(1)


def _warp_matches_golden(hc):
    sc = helpers.golden_warp()
    g = helpers.golden()["warp"]
    depth = synth.to_numpy_u16(sc["depth"])
    warp = synth.warp_arrays(sc)
    v = HcVolume(hc)
    rep = []
    for f in range(sc["n"]):
        d = v.reproject(depth[f], sc["grids"][warp["grid_index"][f]], sc["resolution"], sc["length"], warp["seg"][f], warp["madj"][f])
        assert helpers.digest(d) == g["reprojected_depth"][f]
        rep.append(d)
    v.integrate(np.stack(rep), sc["traj"])
    d = helpers.volume_digest(v)
    assert d["keys"] == g["keys"] and d["sha256"] == g["sha256"]
    assert hc.hc_inside_violations(v.h) == 0 and hc.hc_inside(v.h) > 0
    # the square-root-free "sure" path of k_integrate (voxel_classify), replayed wave by wave: a good share of the (patch, frame)
    # visits take it, and every lane that classifies as sure equals the full update bit for bit
    assert hc.hc_sure_violations(v.h) == 0 and hc.hc_sure(v.h) > 0.25 * hc.hc_visited(v.h), (hc.hc_sure(v.h), hc.hc_visited(v.h))
    print("sure (patch, frame) visits: %d of %d (%.1f %%); with an unsure lane: %d" %
          (hc.hc_sure(v.h), hc.hc_visited(v.h), 100.0 * hc.hc_sure(v.h) / hc.hc_visited(v.h), hc.hc_unsure_pf(v.h)))
    # the FULL verdict (every voxel of the patch updated with tsdf = 1: no projection, no sample): checked voxel by voxel
    assert hc.hc_full_violations(v.h) == 0
    print("full (patch, frame) visits: %d of %d kept (%.1f %%)" % (hc.hc_full(v.h), hc.hc_kept(v.h), 100.0 * hc.hc_full(v.h) / max(hc.hc_kept(v.h), 1)))
    print("register rows of the exact-path visits that need the exact update: %d of %d (%.1f %%)" % (
        hc.hc_exact_rows_needing(v.h), hc.hc_exact_rows(v.h), 100.0 * hc.hc_exact_rows_needing(v.h) / max(hc.hc_exact_rows(v.h), 1)))


def test_device_math_custom_camera_vs_oracle(hc):
    cam = np.array([517.3, 516.5, 318.6, 255.3, 2.5, 1.7], np.float32)
    poses = synth.circle_trajectory(3000)[11::700][:3]
    depth = synth.to_numpy_u16(synth.render_depth(poses, cam=tuple(cam[:4])))
    v, ora = HcVolume(hc, cam), OracleVolume(camera=cam)
    v.integrate(depth, poses)
    for i in range(3):
        ora.Integrate(depth[i], poses[i])
    helpers.assert_volumes_identical(v, ora, "hostcheck/custom camera")
    assert hc.hc_inside_violations(v.h) == 0 and hc.hc_inside(v.h) > 0


def test_band_quotient_core_equals_division_for_every_band_float(hc):
    """The device evaluates (double)sdf / tsdf_trunc_ (TSDFVolume.cpp:88) as q + (x - q c) r with r = RN(1/c) (er_tsdf_math.h
    band_quotient_core).  Checked here for EVERY float voxel_finish can pass, |sdf| <= 0.03f (2 x 1.02e9 values): the float64
    results are identical except for x = -0, which "dp - dist" cannot produce."""
    from concurrent.futures import ThreadPoolExecutor
    hc.hc_band_quotient_check.restype = C.c_long
    hc.hc_band_quotient_check.argtypes = [C.c_uint, C.c_uint, C.POINTER(C.c_uint)]
    top = int(np.float32(0.03).view(np.uint32))
    assert float(np.float32(0.03)) < 0.03 < float(np.nextafter(np.float32(0.03), np.float32(1)))   # the band test's float bound
    nchunk = 16
    edges = np.linspace(0, top + 1, nchunk + 1).astype(np.int64)

    def run(k):
        first = C.c_uint(0)
        bad = hc.hc_band_quotient_check(int(edges[k]), int(edges[k + 1] - 1), C.byref(first))
        return bad, first.value

    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(run, range(nchunk)))
    assert sum(b for b, _ in res) == 1 and res[0] == (1, 0x80000000), res       # only -0


def test_div1000_core_equals_division_on_its_whole_domain(hc):
    """ScaleDepth's x / 1000.f (TSDFVolume.cpp:30) is evaluated on the device as q + (x - 1000 q) r (er_tsdf_math.h div1000_core).
    x = (float)d * lambda is +0, >= 1, +inf or NaN; every such float is compared here (1.07e9 values)."""
    from concurrent.futures import ThreadPoolExecutor
    hc.hc_div1000_check.restype = C.c_long
    hc.hc_div1000_check.argtypes = [C.c_uint, C.c_uint, C.POINTER(C.c_uint)]
    one, last = int(np.float32(1).view(np.uint32)), 0x7FFFFFFF
    edges = np.linspace(one, last + 1, 17).astype(np.int64)

    def run(k):
        first = C.c_uint(0)
        return hc.hc_div1000_check(int(edges[k]), int(edges[k + 1] - 1), C.byref(first)), hex(first.value)

    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(run, range(16)))
    assert all(b == 0 for b, _ in res), res
    first = C.c_uint(0)
    assert hc.hc_div1000_check(0, 0, C.byref(first)) == 0                      # +0
    # and what the domain argument excludes really does differ somewhere below 1 (so the argument is needed)
    assert hc.hc_div1000_check(1, 1 << 20, C.byref(first)) > 0                   # denormal x: the residual underflows


def test_device_math_randomised_configurations_vs_oracle(hc):
    """CPU twin of tests/test_tsdf_gpu.py::test_randomised_configurations_bit_exact for the per-voxel arithmetic and the two
    patch verdicts (culling, "inside"): odd image sizes, off-centre / zero principal points, anisotropic and very short focal
    lengths, cameras anywhere in and around the room (voxels behind and next to the camera plane), holes and salt noise, and a
    scene pushed 40 m away from the origin (large coordinates = large rounding slop in the projection).  Bit-exact against
    the oracle, and every voxel covered by an "inside" verdict re-tested with the full projection."""
    total_inside = 0
    for case in range(6):                                     # 3: cx == 0, 4: very short focal lengths, 5: far from the origin
        rng = np.random.default_rng(7700 + case)
        cols, rows = int(rng.integers(48, 160)), int(rng.integers(40, 120))
        fx, fy = (float(rng.uniform(20, 45)), float(rng.uniform(20, 45))) if case == 4 else (float(rng.uniform(60, 260)), float(rng.uniform(60, 260)))
        cx = 0.0 if case == 3 else float(rng.uniform(0.2, 0.8) * cols)
        cam = np.array([fx, fy, cx, float(rng.uniform(0.2, 0.8) * rows), 2.5, float(rng.uniform(1.0, 3.5))], np.float32)
        n = int(rng.integers(2, 4))
        poses = []
        for _ in range(n):
            P = synth.look_at(tuple(rng.uniform(0.3, 2.7, 3)), tuple(rng.uniform(0.0, 3.0, 3)))
            poses.append(P @ synth.perturbation(int(rng.integers(1 << 30)), 20.0, 0.0))
        poses = np.stack(poses)
        depth = synth.to_numpy_u16(synth.render_depth(poses, cols=cols, rows=rows, cam=tuple(float(c) for c in cam[:4]))).copy()
        depth[rng.random(depth.shape) < 0.05] = 0
        salt = rng.random(depth.shape) < 0.002               # every salt pixel touches a 64^3 unit of its own: keep them few
        depth[salt] = rng.integers(1, 9000, int(salt.sum()), dtype=np.uint16)
        if case == 5:                                         # the same views of a room standing 40 m from the origin
            shift = np.eye(4)
            shift[:3, 3] = (40.0, -35.0, 38.0)
            poses = np.stack([shift @ P for P in poses])
        v, ora = HcVolume(hc, cam, cols, rows), OracleVolume(cols, rows, cam)
        v.integrate(depth, poses)
        for f in range(n):
            ora.Integrate(depth[f], poses[f])
        helpers.assert_volumes_identical(v, ora, "hostcheck fuzz case %d (%dx%d)" % (case, cols, rows))
        assert hc.hc_inside_violations(v.h) == 0, case
        total_inside += hc.hc_inside(v.h)
    assert total_inside > 0


def test_device_reproject_randomised_grids_vs_oracle(hc):
    """CPU twin of the warp half of the GPU fuzz test: CIntegrateApp::Reproject (IntegrateApp.cpp:236-268) through the device
    header -- division-free cube coordinates, fused pixel rounding, their guards and exact fall-backs -- against the oracle for
    randomly deformed control grids, off-centre cameras, holes and salt noise, at the 640 x 480 the reference's bounds assume."""
    for case in range(3):
        rng = np.random.default_rng(9100 + case)
        cam = np.array([float(rng.uniform(400, 650)), float(rng.uniform(400, 650)), 0.0 if case == 2 else float(rng.uniform(200, 440)),
                        float(rng.uniform(150, 330)), 2.5, 2.5], np.float32)
        poses = np.stack([synth.look_at(tuple(rng.uniform(0.5, 2.5, 3)), tuple(rng.uniform(0.2, 2.8, 3))) @
                          synth.perturbation(int(rng.integers(1 << 30)), 15.0, 0.0) for _ in range(2)])
        depth = synth.to_numpy_u16(synth.render_depth(poses, cam=tuple(float(c) for c in cam[:4]))).copy()
        depth[rng.random(depth.shape) < 0.05] = 0
        salt = rng.random(depth.shape) < 0.002
        depth[salt] = rng.integers(1, 12000, int(salt.sum()), dtype=np.uint16)
        res, length = 8, 3.0
        k, j, i = np.meshgrid(np.arange(res + 1), np.arange(res + 1), np.arange(res + 1), indexing="ij")
        base = np.stack([i.ravel(), j.ravel(), k.ravel()], 1) * (length / res)
        grid = (base + rng.normal(0, 0.01 * (1 + 2 * case), base.shape)).astype(np.float32)
        cube = synth.basepose()
        seg = np.stack([cube @ np.linalg.inv(poses[0]) @ P for P in poses])
        madj = np.stack([np.linalg.inv(poses[f]) @ poses[0] @ np.linalg.inv(seg[0]) for f in range(2)])
        v, ora = HcVolume(hc, cam), OracleVolume(640, 480, cam)
        for f in range(2):
            mine = v.reproject(depth[f], grid, res, length, seg[f], madj[f])
            want = ora.Reproject(depth[f], grid, res, np.float32(length), seg[f], madj[f])
            assert np.array_equal(mine, np.asarray(want).reshape(-1)), "case %d frame %d: %d pixels differ" % (
                case, f, int((mine != np.asarray(want).reshape(-1)).sum()))
            assert (mine != 0).sum() > 1000


def test_inside_verdict_stress(hc):
    """The "whole patch inside the image" verdict of patch_may_update on 400 000 random (camera, pose, patch) triples built to
    straddle image borders and the camera plane (focal lengths 20..2000 px, images 32..1280 x 32..960, cameras up to 90 m from
    the origin): wherever it says "inside", all 256 voxels must pass the full voxel_project with the same pixel."""
    from concurrent.futures import ThreadPoolExecutor
    hc.hc_inside_stress.restype = C.c_long
    hc.hc_inside_stress.argtypes = [C.c_ulonglong, C.c_long, C.POINTER(C.c_long)]

    def run(seed):
        n_in = C.c_long(0)
        return hc.hc_inside_stress(seed, 50000, C.byref(n_in)), n_in.value

    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(run, range(1, 9)))
    assert sum(v for v, _ in res) == 0, res
    assert sum(n for _, n in res) > 20000, res                # the verdict does fire: the test is not vacuous


def test_culling_verdict_stress(hc):
    """patch_may_update == false on 80 000 random (camera, pose, patch, depth image) samples built so that every one of its four
    reasons decides both ways (behind the camera, outside the image, no usable depth under the patch, behind the surface by more
    than the truncation): a dropped (patch, frame) pair must not contain a single voxel that voxel_update would change."""
    from concurrent.futures import ThreadPoolExecutor
    hc.hc_cull_stress.restype = C.c_long
    hc.hc_cull_stress.argtypes = [C.c_ulonglong, C.c_long, C.POINTER(C.c_long)]

    def run(seed):
        n_dead = C.c_long(0)
        return hc.hc_cull_stress(seed, 10000, C.byref(n_dead)), n_dead.value

    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(run, range(1, 9)))
    assert sum(w for w, _ in res) == 0, res
    dead = sum(n for _, n in res)
    assert 8000 < dead < 72000, res                          # it decides both ways


def test_touch_key_stress(hc):
    """The guarded division-free unit key of k_prepare against the reference's expression: random cameras, poses up to 150 m
    out (keys of -1 occur), the whole 16-bit depth range, and points placed within a few float64 ulps of a unit boundary."""
    n_out = C.c_long(0)
    assert hc.hc_touch_key_stress(5, 3000000, C.byref(n_out)) == 0
    assert 100000 < n_out.value < 2500000, n_out.value


def test_full_verdict_stress(hc):
    """patch_may_update_box's third verdict on its own: random cameras, poses, patches (box / square / strip) and depth images
    whose surface lies around and behind the patch, with and without holes; wherever it says "full", every voxel of the patch
    must be updated by the full voxel_update exactly like the tsdf = 1 shortcut, from fresh, S == 1 (also W = 2^24) and
    arbitrary voxel states."""
    from concurrent.futures import ThreadPoolExecutor

    def run(seed):
        n_full = C.c_long(0)
        return hc.hc_full_stress(seed, 2500, C.byref(n_full)), n_full.value
    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(run, range(31, 39)))
    assert sum(w for w, _ in res) == 0, res
    assert sum(n for _, n in res) > 2000, res                 # it fires often enough to mean something


def test_sure_classification_stress(hc):
    """voxel_classify on its own: scaled depths up to the 64 m bound, squared distances on and within a few ulps of both decision
    thresholds and at dist = dp +- trunc; wherever a lane classifies as sure (behind, or free with S == 1 / W == 0) the
    shortcut must reproduce voxel_finish_d2 bit for bit from fresh, S == 1 (also W = 2^24) and arbitrary voxel states."""
    n_sure = C.c_long(0)
    assert hc.hc_sure_stress(21, 4000000, C.byref(n_sure)) == 0
    assert n_sure.value > 2000000, n_sure.value


@pytest.mark.parametrize("ulps", [1, -1, 2, -2])
def test_verdicts_survive_a_hardware_reciprocal(ulps, tmp_path):
    """The device build (since round 2; -DER_CULL_IEEE_DIV restores the divisions) evaluates the corner projections of patch_may_update with v_rcp_f32
    instead of an IEEE division.  Before it is enabled: the same header compiled for the host with a reciprocal that is off by
    1 or 2 ulps in either direction, under both verdict stress tests."""
    src = os.path.join(HERE, "hostcheck", "tsdf_hostcheck.cpp")
    out = str(tmp_path / "libhc_sim.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DER_FAST_CULL_HOSTSIM=(%d)" % ulps, src, "-o", out], check=True)
    L = C.CDLL(out)
    for fn in (L.hc_inside_stress, L.hc_cull_stress):
        fn.restype = C.c_long
        fn.argtypes = [C.c_ulonglong, C.c_long, C.POINTER(C.c_long)]
    n_in, n_dead = C.c_long(0), C.c_long(0)
    assert L.hc_inside_stress(11, 100000, C.byref(n_in)) == 0 and n_in.value > 5000
    assert L.hc_cull_stress(12, 15000, C.byref(n_dead)) == 0 and n_dead.value > 2000
