"""ctypes binding of liber_hip.so -- the C ABI declared in include/er_hip.h.

There is no CPU fallback: if the shared library is missing, or a constructor cannot find a HIP
device, the call raises.  Nothing under oracle/ is imported from here.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ER_HIP_LIB selects another build of the same library (A/B measurements of kernel variants); never a fallback.
LIB_PATH = os.environ.get("ER_HIP_LIB") or os.path.join(_HERE, "liber_hip.so")

# Every symbol include/er_hip.h declares (tests/test_abi.py checks header == this list == the .so).
SYMBOLS = [
    "er_last_error", "er_device_count", "er_abi_version", "er_request_hw_queues", "er_host_alloc", "er_host_free", "er_host_copy_h2d", "er_device_alloc", "er_device_free", "er_device_copy_d2h",
    "er_tsdf_create", "er_tsdf_destroy", "er_tsdf_set_stream", "er_tsdf_synchronize",
    "er_tsdf_wait_event", "er_tsdf_reset", "er_tsdf_status", "er_tsdf_set_unit_shard", "er_unit_owner",
    "er_tsdf_scale_depth", "er_tsdf_reproject", "er_tsdf_integrate", "er_tsdf_integrate_frames",
    "er_tsdf_unit_count", "er_tsdf_unit_keys", "er_tsdf_read_unit", "er_tsdf_sum_weight",
    "er_tsdf_extract_world", "er_tsdf_extract_surface", "er_tsdf_extract_mesh", "er_mc_table", "er_tsdf_export_weighted", "er_tsdf_import_weighted",
    "er_tsdf_export_raw", "er_tsdf_import_raw",
    "er_tsdf_band_sizes", "er_tsdf_export_band", "er_tsdf_merge_band", "er_tsdf_import_band", "er_tsdf_drop_units",
    "er_tsdf_set_profiling", "er_tsdf_get_profile",
    "er_comm_unique_id", "er_comm_create", "er_comm_create_local", "er_comm_create_loopback", "er_comm_destroy", "er_comm_rank", "er_comm_world",
    "er_tsdf_allreduce", "er_comm_merge_stats", "er_comm_merge_stats_owner", "er_frame_block",
    "er_cloud_create", "er_cloud_create_batch", "er_cloud_destroy", "er_cloud_size",
    "er_icp_count_inliers", "er_icp_align", "er_find_correspondence",
    "er_icp_count_inliers_batch", "er_icp_align_batch", "er_find_correspondence_batch", "er_icp_release_workspaces", "er_registration_batch", "er_ransac_fitness_batch", "er_ransac_inliers",
    "er_fopt_create", "er_fopt_destroy", "er_fopt_set_cloud", "er_fopt_cloud_size", "er_fopt_get_points", "er_fopt_update_pose",
    "er_fopt_update_point_pn", "er_fopt_set_correspondences", "er_fopt_set_correspondences_dev", "er_fopt_group_count", "er_fopt_group_info", "er_fopt_update_normals", "er_fopt_assemble_rigid", "er_fopt_assemble_slac",
    "er_fopt_assemble_nonrigid", "er_fopt_factor_slac", "er_fopt_factor_nonrigid", "er_fopt_solve", "er_fopt_debug_shift_diagonal",
]


class ErWarp(C.Structure):
    """struct er_warp (include/er_hip.h)."""
    _fields_ = [
        ("ctr", C.POINTER(C.c_float)),
        ("num_grids", C.c_int),
        ("resolution", C.c_int),
        ("length", C.c_float),
        ("grid_index", C.POINTER(C.c_int)),
        ("seg", C.POINTER(C.c_double)),
        ("madj", C.POINTER(C.c_double)),
    ]


class ErError(RuntimeError):
    pass


_lib = None


def lib():
    """Load liber_hip.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ErError(
            "liber_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C elasticreconstruction_amd/csrc`; there is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, ip, dp, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_float)
    u16p = C.POINTER(C.c_uint16)
    L.er_last_error.restype = C.c_char_p
    L.er_last_error.argtypes = []
    L.er_device_count.argtypes = []
    L.er_abi_version.argtypes = []
    L.er_host_alloc.restype = vp
    L.er_host_alloc.argtypes = [C.c_size_t]
    L.er_host_free.argtypes = [vp]
    L.er_host_copy_h2d.argtypes = [vp, vp, C.c_size_t]
    L.er_device_alloc.restype = vp
    L.er_device_alloc.argtypes = [C.c_size_t, C.c_int]
    L.er_device_free.argtypes = [vp]
    L.er_device_copy_d2h.argtypes = [vp, vp, C.c_size_t]
    L.er_tsdf_create.argtypes = [C.c_int, C.c_int, fp, C.c_int, C.c_int, C.POINTER(vp)]
    L.er_tsdf_destroy.argtypes = [vp]
    L.er_tsdf_set_stream.argtypes = [vp, vp]
    L.er_tsdf_synchronize.argtypes = [vp]
    L.er_tsdf_wait_event.argtypes = [vp, vp]
    L.er_tsdf_reset.argtypes = [vp]
    L.er_tsdf_status.argtypes = [vp, ip, C.POINTER(C.c_long)]
    L.er_tsdf_set_unit_shard.argtypes = [vp, C.c_int, C.c_int]
    L.er_unit_owner.argtypes = [C.c_int, C.c_int]
    L.er_tsdf_scale_depth.argtypes = [vp, vp, vp]
    L.er_tsdf_reproject.argtypes = [vp, vp, vp, C.c_int, C.c_float, vp, vp]
    L.er_tsdf_integrate.argtypes = [vp, vp, vp]
    L.er_tsdf_integrate_frames.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.POINTER(ErWarp)]
    L.er_tsdf_unit_count.argtypes = [vp, ip]
    L.er_tsdf_unit_keys.argtypes = [vp, vp]
    L.er_tsdf_read_unit.argtypes = [vp, C.c_int, vp, vp]
    L.er_tsdf_sum_weight.argtypes = [vp, dp]
    L.er_tsdf_extract_world.argtypes = [vp, vp, C.c_long, C.POINTER(C.c_long)]
    L.er_tsdf_extract_surface.argtypes = [vp, vp, C.c_long, C.POINTER(C.c_long)]
    L.er_tsdf_extract_mesh.argtypes = [vp, vp, C.c_long, C.POINTER(C.c_long)]
    L.er_mc_table.argtypes = [vp]
    L.er_request_hw_queues.argtypes = [C.c_int]
    L.er_tsdf_export_weighted.argtypes = [vp, vp, C.c_int, vp]
    L.er_tsdf_import_weighted.argtypes = [vp, vp, C.c_int, vp]
    L.er_tsdf_export_raw.argtypes = [vp, vp, C.c_int, vp]
    L.er_tsdf_import_raw.argtypes = [vp, vp, C.c_int, vp]
    L.er_comm_unique_id.argtypes = [vp]
    L.er_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.er_comm_create_local.argtypes = [C.c_int, vp, vp]
    L.er_comm_create_loopback.argtypes = [C.c_int, C.c_int, vp]
    L.er_comm_destroy.argtypes = [vp]
    L.er_comm_rank.argtypes = [vp]
    L.er_comm_world.argtypes = [vp]
    L.er_tsdf_allreduce.argtypes = [vp, vp, C.c_int, ip]
    L.er_comm_merge_stats.argtypes = [vp, vp]
    L.er_comm_merge_stats_owner.argtypes = [vp, vp]
    L.er_tsdf_band_sizes.argtypes = [vp, vp, C.c_int, vp]
    L.er_tsdf_export_band.argtypes = [vp, vp, vp, C.c_int, vp]
    L.er_tsdf_merge_band.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.er_tsdf_import_band.argtypes = [vp, vp, C.c_int, vp]
    L.er_tsdf_drop_units.argtypes = [vp, vp, C.c_int]
    L.er_frame_block.restype = None
    L.er_frame_block.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip]
    L.er_tsdf_set_profiling.argtypes = [vp, C.c_int]
    L.er_tsdf_get_profile.argtypes = [vp, dp, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    if hasattr(L, "er_cloud_create"):
        L.er_cloud_create.argtypes = [vp, vp, C.c_int, C.c_float, C.c_int, C.POINTER(vp)]
        L.er_cloud_create_batch.argtypes = [C.c_int, vp, vp, vp, C.c_float, C.c_int, vp]
        L.er_cloud_destroy.argtypes = [vp]
        L.er_cloud_size.argtypes = [vp]
        L.er_icp_count_inliers.argtypes = [vp, vp, vp, C.c_double, ip]
        L.er_icp_align.argtypes = [vp, vp, vp, C.c_double, C.c_int, C.c_double, C.c_int, vp, ip, ip, dp]
        L.er_find_correspondence.argtypes = [vp, vp, vp, C.c_double, C.c_double, vp, C.c_int, ip, vp]
        L.er_icp_count_inliers_batch.argtypes = [C.c_int, vp, vp, vp, C.c_double, vp]
        L.er_icp_align_batch.argtypes = [C.c_int, vp, vp, vp, C.c_double, C.c_int, C.c_double, C.c_int, vp, vp, vp, vp]
        L.er_find_correspondence_batch.argtypes = [C.c_int, vp, vp, vp, C.c_double, C.c_double, vp, vp, vp, vp]
        L.er_icp_release_workspaces.argtypes = []
        L.er_registration_batch.argtypes = [C.c_int, vp, vp, vp, C.c_double, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, C.c_double, C.c_double,
                                            vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.er_ransac_fitness_batch.argtypes = [vp, vp, C.c_int, vp, C.c_float, vp, vp]
        L.er_ransac_inliers.argtypes = [vp, vp, vp, C.c_float, vp, C.c_int, ip, vp, vp, vp]
    if hasattr(L, "er_fopt_create"):
        L.er_fopt_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(vp)]
        L.er_fopt_destroy.argtypes = [vp]
        L.er_fopt_set_cloud.argtypes = [vp, C.c_int, vp, vp, C.c_int, ip]
        L.er_fopt_cloud_size.argtypes = [vp, C.c_int]
        L.er_fopt_get_points.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
        L.er_fopt_update_pose.argtypes = [vp, C.c_int, vp]
        L.er_fopt_update_point_pn.argtypes = [vp, C.c_int, vp]
        L.er_fopt_set_correspondences.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.er_fopt_set_correspondences_dev.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.er_fopt_group_count.argtypes = [vp]
        L.er_fopt_assemble_rigid.argtypes = [vp, vp, vp, vp]
        L.er_fopt_assemble_slac.argtypes = [vp, vp, vp, vp, vp]
        L.er_fopt_group_info.argtypes = [vp, vp]
        L.er_fopt_update_normals.argtypes = [vp, C.c_int, vp]
        L.er_fopt_assemble_nonrigid.argtypes = [vp, C.c_double, vp, vp]
        L.er_fopt_factor_slac.argtypes = [vp, vp, C.c_double, vp, vp]
        L.er_fopt_factor_nonrigid.argtypes = [vp, C.c_double]
        L.er_fopt_solve.argtypes = [vp, vp, C.c_int, vp]
        L.er_fopt_debug_shift_diagonal.argtypes = [vp, C.c_long, C.c_double]
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        msg = lib().er_last_error()
        raise ErError("%s failed: %s" % (what, msg.decode("utf-8", "replace") if msg else "unknown error"))


class PinnedArena:
    """Grow-only block of page-locked host memory (er_host_alloc) handed out as numpy views.  Results written into
    it arrive by asynchronous device-to-host copies with no staging pass; take(...) views stay valid until the next
    reset() of the same arena."""

    def __init__(self):
        self._lib, self._p, self._cap, self._off, self._old = lib(), None, 0, 0, []

    def reset(self, need_bytes):
        self._off = 0
        for p in self._old:
            self._lib.er_host_free(p)
        self._old = []
        if need_bytes > self._cap:
            if self._p:
                self._lib.er_host_free(self._p)
            self._cap = int(need_bytes * 1.25) + 4096
            self._p = self._lib.er_host_alloc(self._cap)
            if not self._p:
                self._cap = 0
                raise ErError("er_host_alloc: " + self._lib.er_last_error().decode())

    def take(self, shape, dtype):
        import numpy as np
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        off = (self._off + 4095) & ~4095
        assert off + n <= self._cap, "PinnedArena.reset() was sized too small"
        self._off = off + n
        buf = (C.c_char * max(n, 1)).from_address(self._p + off)
        return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)

    def close(self):
        if self._p:
            self._lib.er_host_free(self._p)
            self._p, self._cap = None, 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ptr(a):
    """numpy array -> void* (the array must stay alive for the duration of the call)."""
    return a.ctypes.data_as(C.c_void_p)
