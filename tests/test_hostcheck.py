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
    L.hc_unit_keys.argtypes = [vp, vp]
    L.hc_read_unit.argtypes = [vp, C.c_int, vp, vp]
    return L


class HcVolume:
    def __init__(self, L, cam=None):
        self.L = L
        self.cam = np.array([525.0, 525.0, 319.5, 239.5, 2.5, 2.5], np.float32) if cam is None else np.asarray(cam, np.float32)
        self.h = vp(L.hc_create(640, 480, self.cam.ctypes.data_as(vp)))

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


def test_device_math_warp_matches_golden(hc):
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


def test_device_math_custom_camera_vs_oracle(hc):
    cam = np.array([517.3, 516.5, 318.6, 255.3, 2.5, 1.7], np.float32)
    poses = synth.circle_trajectory(3000)[11::700][:3]
    depth = synth.to_numpy_u16(synth.render_depth(poses, cam=tuple(cam[:4])))
    v, ora = HcVolume(hc, cam), OracleVolume(camera=cam)
    v.integrate(depth, poses)
    for i in range(3):
        ora.Integrate(depth[i], poses[i])
    helpers.assert_volumes_identical(v, ora, "hostcheck/custom camera")
