"""Host-side mirror of the reference's Integrate program over the C ABI (include/er_hip.h).

  TSDFVolume    <->  Integrate/TSDFVolume.h:14-69   (ScaleDepth, Integrate, SaveWorld, data_)
  IntegrateApp  <->  Integrate/IntegrateApp.h:33-94 (Init, Execute gating, Reproject) -- same member
                     names as the reference so parity tests read like the reference's own flow.

All arithmetic runs in the HIP kernels of liber_hip.so; this file only marshals buffers and
reproduces the reference's per-frame control flow (frame ids are 1-based, IntegrateApp.cpp:185).
"""
import ctypes as C

import numpy as np

from . import _ffi
from . import formats

UNIT_VOX = 64 * 64 * 64


def mat4_mul(A, B):
    """Row-major 4x4 float64 product in Eigen's coefficient order ((a0*b0 + a1*b1) + a2*b2) + a3*b3."""
    A = np.asarray(A, np.float64)
    B = np.asarray(B, np.float64)
    out = np.empty((4, 4), np.float64)
    for r in range(4):
        for c in range(4):
            out[r, c] = ((A[r, 0] * B[0, c] + A[r, 1] * B[1, c]) + A[r, 2] * B[2, c]) + A[r, 3] * B[3, c]
    return out


class TSDFVolume:
    """TSDFVolume (TSDFVolume.h:14-69) resident in HBM on one GPU."""

    def __init__(self, cols=640, rows=480, camera=None, max_units=1024, device=0):
        self._lib = _ffi.lib()
        self.cols_, self.rows_ = int(cols), int(rows)
        cam = None if camera is None else np.ascontiguousarray(camera, dtype=np.float32)
        self.camera_ = formats.load_camera(None) if cam is None else cam
        self.max_units = int(max_units)
        self.device = int(device)
        h = C.c_void_p()
        _ffi.check(self._lib.er_tsdf_create(self.cols_, self.rows_,
                                            self.camera_.ctypes.data_as(C.POINTER(C.c_float)),
                                            self.max_units, self.device, C.byref(h)), "er_tsdf_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.er_tsdf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- stream / sync -------------------------------------------------------------------------
    def set_stream(self, hip_stream_ptr):
        _ffi.check(self._lib.er_tsdf_set_stream(self._h, C.c_void_p(hip_stream_ptr)), "er_tsdf_set_stream")

    def synchronize(self):
        _ffi.check(self._lib.er_tsdf_synchronize(self._h), "er_tsdf_synchronize")

    def wait_event(self, hip_event_ptr):
        """The next IntegrateFrames call's pre-pass waits for this hipEvent_t (device depth produced asynchronously)."""
        _ffi.check(self._lib.er_tsdf_wait_event(self._h, C.c_void_p(hip_event_ptr)), "er_tsdf_wait_event")

    def reset(self):
        """data_.clear(): an empty volume in the same buffers."""
        _ffi.check(self._lib.er_tsdf_reset(self._h), "er_tsdf_reset")

    def set_unit_shard(self, rank, world):
        """Unit-shard mode (bit-exact): this volume only owns the units with er_unit_owner(key, world) == rank."""
        _ffi.check(self._lib.er_tsdf_set_unit_shard(self._h, int(rank), int(world)), "er_tsdf_set_unit_shard")

    def status(self):
        """(flags, out_of_range_pixels): sticky overflow flags (ER_STATUS_*) and the pixels skipped beyond +-96 m; a poll."""
        f, n = C.c_int(0), C.c_long(0)
        _ffi.check(self._lib.er_tsdf_status(self._h, C.byref(f), C.byref(n)), "er_tsdf_status")
        return f.value, n.value

    # -- numeric core ---------------------------------------------------------------------------
    def ScaleDepth(self, depth):
        """TSDFVolume::ScaleDepth (TSDFVolume.cpp:19-36)."""
        d = np.ascontiguousarray(depth, dtype=np.uint16).reshape(-1)
        assert d.size == self.cols_ * self.rows_
        out = np.empty(d.size, dtype=np.float32)
        _ffi.check(self._lib.er_tsdf_scale_depth(self._h, _ffi.ptr(d), _ffi.ptr(out)), "er_tsdf_scale_depth")
        return out

    def Reproject(self, depth, ctr, resolution, length, seg, madj):
        """Pixel loop of CIntegrateApp::Reproject (IntegrateApp.cpp:236-268) for one frame; returns the new depth."""
        d = np.array(depth, dtype=np.uint16).reshape(-1)
        g = np.ascontiguousarray(ctr, dtype=np.float32).reshape(-1)
        s = np.ascontiguousarray(seg, dtype=np.float64).reshape(16)
        m = np.ascontiguousarray(madj, dtype=np.float64).reshape(16)
        _ffi.check(self._lib.er_tsdf_reproject(self._h, _ffi.ptr(d), _ffi.ptr(g), int(resolution), C.c_float(length),
                                               _ffi.ptr(s), _ffi.ptr(m)), "er_tsdf_reproject")
        return d

    def Integrate(self, depth, transformation):
        """ScaleDepth + TSDFVolume::Integrate for one frame (IntegrateApp.cpp:224-225)."""
        d = np.ascontiguousarray(depth, dtype=np.uint16).reshape(-1)
        T = np.ascontiguousarray(transformation, dtype=np.float64).reshape(16)
        _ffi.check(self._lib.er_tsdf_integrate(self._h, _ffi.ptr(d), _ffi.ptr(T)), "er_tsdf_integrate")

    def IntegrateFrames(self, depth, transformations, warp=None, device_ptr=None):
        """n frames in order.  depth: uint16[n, rows*cols] on the host, or device_ptr = address of the same
        layout in HBM.  warp = dict(ctr=float32[num, verts, 3], resolution, length, grid_index=int[n],
        seg=float64[n,4,4], madj=float64[n,4,4]) or None."""
        T = np.ascontiguousarray(transformations, dtype=np.float64).reshape(-1, 16)
        n = T.shape[0]
        w_ref = None
        keep = []
        if warp is not None:
            ctr = np.ascontiguousarray(warp["ctr"], dtype=np.float32)
            gi = np.ascontiguousarray(warp["grid_index"], dtype=np.int32).reshape(-1)
            seg = np.ascontiguousarray(warp["seg"], dtype=np.float64).reshape(-1, 16)
            madj = np.ascontiguousarray(warp["madj"], dtype=np.float64).reshape(-1, 16)
            assert gi.size == n and seg.shape[0] == n and madj.shape[0] == n
            res = int(warp["resolution"])
            num = ctr.size // (3 * (res + 1) ** 3)
            w = _ffi.ErWarp(ctr.ctypes.data_as(C.POINTER(C.c_float)), num, res, C.c_float(warp["length"]),
                            gi.ctypes.data_as(C.POINTER(C.c_int)), seg.ctypes.data_as(C.POINTER(C.c_double)),
                            madj.ctypes.data_as(C.POINTER(C.c_double)))
            keep = [ctr, gi, seg, madj, w]
            w_ref = C.byref(w)
        if device_ptr is not None:
            _ffi.check(self._lib.er_tsdf_integrate_frames(self._h, n, C.c_void_p(device_ptr), 1, _ffi.ptr(T), w_ref),
                       "er_tsdf_integrate_frames")
        else:
            d = np.ascontiguousarray(depth, dtype=np.uint16).reshape(n, -1)
            assert d.shape[1] == self.cols_ * self.rows_
            _ffi.check(self._lib.er_tsdf_integrate_frames(self._h, n, _ffi.ptr(d), 0, _ffi.ptr(T), w_ref),
                       "er_tsdf_integrate_frames")
        del keep

    # -- data_ access ---------------------------------------------------------------------------
    def unit_count(self):
        n = C.c_int(0)
        _ffi.check(self._lib.er_tsdf_unit_count(self._h, C.byref(n)), "er_tsdf_unit_count")
        return n.value

    def unit_keys(self):
        n = self.unit_count()
        keys = np.empty(n, dtype=np.int32)
        if n:
            _ffi.check(self._lib.er_tsdf_unit_keys(self._h, _ffi.ptr(keys)), "er_tsdf_unit_keys")
        return keys

    def read_unit(self, key):
        sdf = np.empty(UNIT_VOX, dtype=np.float32)
        w = np.empty(UNIT_VOX, dtype=np.float32)
        _ffi.check(self._lib.er_tsdf_read_unit(self._h, int(key), _ffi.ptr(sdf), _ffi.ptr(w)), "er_tsdf_read_unit")
        return sdf, w

    def sum_weight(self):
        s = C.c_double(0)
        _ffi.check(self._lib.er_tsdf_sum_weight(self._h, C.byref(s)), "er_tsdf_sum_weight")
        return s.value

    def extract_world(self):
        """SaveWorld's point list (TSDFVolume.cpp:104-132) as float32[n, 4] = x y z intensity."""
        n = C.c_long(0)
        _ffi.check(self._lib.er_tsdf_extract_world(self._h, None, 0, C.byref(n)), "er_tsdf_extract_world")
        out = np.empty((n.value, 4), dtype=np.float32)
        if n.value:
            _ffi.check(self._lib.er_tsdf_extract_world(self._h, _ffi.ptr(out), n.value, C.byref(n)), "er_tsdf_extract_world")
        return out

    def extract_surface(self):
        """Zero-crossing points of the volume (er_tsdf_extract_surface) as float32[n, 4] = x y z axis, metres."""
        n = C.c_long(0)
        _ffi.check(self._lib.er_tsdf_extract_surface(self._h, None, 0, C.byref(n)), "er_tsdf_extract_surface")
        out = np.empty((n.value, 4), dtype=np.float32)
        if n.value:
            _ffi.check(self._lib.er_tsdf_extract_surface(self._h, _ffi.ptr(out), n.value, C.byref(n)), "er_tsdf_extract_surface")
        return out

    def extract_mesh(self):
        """Marching-cubes triangles of the volume (er_tsdf_extract_mesh) as float32[n, 3, 3] = triangle, vertex, xyz in metres."""
        n = C.c_long(0)
        _ffi.check(self._lib.er_tsdf_extract_mesh(self._h, None, 0, C.byref(n)), "er_tsdf_extract_mesh")
        out = np.empty((n.value, 3, 3), dtype=np.float32)
        if n.value:
            _ffi.check(self._lib.er_tsdf_extract_mesh(self._h, _ffi.ptr(out), n.value, C.byref(n)), "er_tsdf_extract_mesh")
        return out

    def SaveWorld(self, filename):
        pts = self.extract_world()
        formats.save_pcd_xyzi(filename, pts)
        return pts.shape[0]

    # -- multi-GPU frame split ----------------------------------------------------
    def export_weighted(self, keys, dev_ptr):
        k = np.ascontiguousarray(keys, dtype=np.int32)
        _ffi.check(self._lib.er_tsdf_export_weighted(self._h, _ffi.ptr(k), k.size, C.c_void_p(dev_ptr)), "er_tsdf_export_weighted")

    def import_weighted(self, keys, dev_ptr):
        k = np.ascontiguousarray(keys, dtype=np.int32)
        _ffi.check(self._lib.er_tsdf_import_weighted(self._h, _ffi.ptr(k), k.size, C.c_void_p(dev_ptr)), "er_tsdf_import_weighted")

    def export_raw(self, keys, dev_ptr):
        """[key][sdf | weight] planes, bit for bit: how a unit only one GPU touched travels (er_tsdf_export_raw)."""
        k = np.ascontiguousarray(keys, dtype=np.int32)
        _ffi.check(self._lib.er_tsdf_export_raw(self._h, _ffi.ptr(k), k.size, C.c_void_p(dev_ptr)), "er_tsdf_export_raw")

    def import_raw(self, keys, dev_ptr):
        k = np.ascontiguousarray(keys, dtype=np.int32)
        _ffi.check(self._lib.er_tsdf_import_raw(self._h, _ffi.ptr(k), k.size, C.c_void_p(dev_ptr)), "er_tsdf_import_raw")

    # band records (round 6, the owner merge): a unit as its observed voxels only -- include/er_hip.h
    def band_sizes(self, keys):
        """Record size in 32-bit words of each of the given units (er_tsdf_band_sizes)."""
        k = np.ascontiguousarray(keys, dtype=np.int32)
        out = np.zeros(k.size, np.int32)
        _ffi.check(self._lib.er_tsdf_band_sizes(self._h, _ffi.ptr(k), k.size, _ffi.ptr(out)), "er_tsdf_band_sizes")
        return out

    def export_band(self, keys, counts, dev_ptr):
        k, c = np.ascontiguousarray(keys, dtype=np.int32), np.ascontiguousarray(counts, dtype=np.int32)
        _ffi.check(self._lib.er_tsdf_export_band(self._h, _ffi.ptr(k), _ffi.ptr(c), k.size, C.c_void_p(dev_ptr)), "er_tsdf_export_band")

    def import_band(self, keys, rec_ptrs):
        k = np.ascontiguousarray(keys, dtype=np.int32)
        r = (C.c_void_p * k.size)(*[int(p) for p in rec_ptrs])
        _ffi.check(self._lib.er_tsdf_import_band(self._h, _ffi.ptr(k), k.size, r), "er_tsdf_import_band")

    def merge_band(self, keys, srcs, self_pos):
        """srcs[u] = device pointers of the other touchers' records of unit keys[u] in rank order; self_pos[u] of them come before this volume's own voxels."""
        k = np.ascontiguousarray(keys, dtype=np.int32)
        ns = np.asarray([len(x) for x in srcs], np.int32)
        sp = np.ascontiguousarray(self_pos, dtype=np.int32)
        r = (C.c_void_p * (16 * k.size))()
        for u, x in enumerate(srcs):
            for q, ptr in enumerate(x):
                r[16 * u + q] = int(ptr)
        _ffi.check(self._lib.er_tsdf_merge_band(self._h, _ffi.ptr(k), k.size, _ffi.ptr(ns), _ffi.ptr(sp), r), "er_tsdf_merge_band")

    def drop_units(self, keys):
        k = np.ascontiguousarray(keys, dtype=np.int32)
        _ffi.check(self._lib.er_tsdf_drop_units(self._h, _ffi.ptr(k), k.size), "er_tsdf_drop_units")

    # -- profiling --------------------------------------------------------------------------------
    def set_profiling(self, enable):
        """True / 1: time every k_integrate launch; n > 1: every n-th launch; False / 0: off."""
        _ffi.check(self._lib.er_tsdf_set_profiling(self._h, int(enable)), "er_tsdf_set_profiling")

    def get_profile(self):
        ms, ln, fr, uv = C.c_double(0), C.c_long(0), C.c_long(0), C.c_long(0)
        _ffi.check(self._lib.er_tsdf_get_profile(self._h, C.byref(ms), C.byref(ln), C.byref(fr), C.byref(uv)), "er_tsdf_get_profile")
        return {"integrate_ms": ms.value, "launches": ln.value, "frames": fr.value, "unit_visits": uv.value}


class IntegrateApp:
    """CIntegrateApp (IntegrateApp.h:33-94) without the OpenNI grabber: frames are fed by the caller.

    Member names follow the reference.  Execute() reproduces the per-frame gating of
    IntegrateApp.cpp:190-226 exactly (including its quirks); frames that pass
    the gates are queued and flushed to the GPU in batches -- the results equal frame-by-frame
    execution because the device processes a batch in frame order per voxel.
    """

    def __init__(self, cols=640, rows=480, max_units=1024, device=0, batch=64):
        self.cols_, self.rows_ = cols, rows
        self.traj_filename_ = ""
        self.pose_filename_ = ""
        self.seg_filename_ = ""
        self.camera_filename_ = ""
        self.ctr_filename_ = ""
        self.pcd_filename_ = "world.pcd"
        self.ctr_num_ = 0
        self.ctr_resolution_ = 8
        self.ctr_interval_ = 50
        self.ctr_length_ = 3.0
        self.start_from_ = -1
        self.end_at_ = 100000000
        self.exit_ = False
        self.frame_id_ = 0
        self.traj_ = []
        self.seg_traj_ = []
        self.pose_traj_ = []
        self.grids_ = None
        self.volume_ = None
        self._max_units, self._device, self._batch = max_units, device, int(batch)
        self._q_depth, self._q_T, self._q_gi, self._q_seg, self._q_madj = [], [], [], [], []
        self.frames_integrated = 0

    def Init(self):
        """CIntegrateApp::Init (IntegrateApp.cpp:43-79)."""
        import os
        camera = formats.load_camera(self.camera_filename_ if os.path.exists(self.camera_filename_ or "") else None)
        self.volume_ = TSDFVolume(self.cols_, self.rows_, camera, self._max_units, self._device)
        if self.ctr_num_ > 0 and os.path.exists(self.ctr_filename_ or "") and os.path.exists(self.seg_filename_ or ""):
            self.grids_ = formats.load_ctr(self.ctr_filename_, self.ctr_num_, self.ctr_resolution_)
        else:
            self.ctr_num_ = 0
        if os.path.exists(self.traj_filename_ or ""):
            self.traj_ = formats.load_log(self.traj_filename_)
        if os.path.exists(self.seg_filename_ or ""):
            self.seg_traj_ = formats.load_log(self.seg_filename_)
            if os.path.exists(self.pose_filename_ or ""):
                self.pose_traj_ = formats.load_log(self.pose_filename_)
                self.traj_ = []
                for i in range(len(self.pose_traj_)):
                    for j in range(self.ctr_interval_):
                        idx = i * self.ctr_interval_ + j
                        self.traj_.append(formats.FramedTransformation(
                            idx, idx, idx + 1, mat4_mul(self.pose_traj_[i].T, self.seg_traj_[idx].T)))

    def Execute(self, frame_id, depth):
        """One main-loop turn with data (IntegrateApp.cpp:190-226); frame_id is 1-based."""
        self.frame_id_ = frame_id
        if self.traj_[frame_id - 1].frame == -1:
            return
        if frame_id >= len(self.traj_):
            self.exit_ = True
            return
        if frame_id < self.start_from_ or frame_id > self.end_at_:
            if frame_id > self.end_at_:
                self.exit_ = True
            return
        T = self.traj_[frame_id - 1].T
        if self.ctr_num_ > 0:
            if frame_id > self.ctr_interval_ * self.ctr_num_:          # Reproject, IntegrateApp.cpp:230-233
                self.exit_ = True
                return
            chunk = (frame_id - 1) // self.ctr_interval_
            madj = mat4_mul(mat4_mul(_inverse(T), self.traj_[0].T), _inverse(self.seg_traj_[0].T))
            self._q_gi.append(chunk)
            self._q_seg.append(self.seg_traj_[frame_id - 1].T.copy())
            self._q_madj.append(madj)
        self._q_depth.append(np.ascontiguousarray(depth, dtype=np.uint16).reshape(-1))
        self._q_T.append(T.copy())
        if len(self._q_depth) >= self._batch:
            self.Flush()

    def Flush(self):
        if not self._q_depth:
            return
        warp = None
        if self.ctr_num_ > 0:
            warp = dict(ctr=self.grids_, resolution=self.ctr_resolution_, length=np.float32(self.ctr_length_),
                        grid_index=np.array(self._q_gi, np.int32), seg=np.array(self._q_seg), madj=np.array(self._q_madj))
        self.volume_.IntegrateFrames(np.stack(self._q_depth), np.array(self._q_T), warp)
        self.frames_integrated += len(self._q_depth)
        self._q_depth, self._q_T, self._q_gi, self._q_seg, self._q_madj = [], [], [], [], []

    def Finish(self, save=True):
        self.Flush()
        if save:
            return self.volume_.SaveWorld(self.pcd_filename_)
        return None


def _inverse(T):
    """4x4 float64 inverse by cofactors -- same expansion as csrc/er_common.cpp so the host programs and
    this mirror hand identical matrices to the kernels."""
    m = np.asarray(T, np.float64).reshape(16)
    inv = np.empty(16, np.float64)
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10]
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10]
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9]
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9]
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10]
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10]
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9]
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9]
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6]
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6]
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5]
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5]
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6]
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6]
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5]
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5]
    det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12]
    det = 1.0 / det
    return (inv * det).reshape(4, 4)


def reproject_matrix(traj_f, traj_0, seg_0):
    """IntegrateApp.cpp:243: traj[f-1].inverse() * traj[0] * seg[0].inverse()."""
    return mat4_mul(mat4_mul(_inverse(traj_f), traj_0), _inverse(seg_0))
