"""ctypes access to the CPU checkers -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The
product package (elasticreconstruction_amd/) never does.

  OracleVolume   oracle/tsdf_oracle.c            scalar C restatement of path A ("port")
  RefApp         oracle/_ref/libref_tsdf*.so     the reference's own CIntegrateApp / TSDFVolume,
                                                 compiled unmodified from /root/reference ("reference")
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
UNIT_VOX = 64 * 64 * 64
_vp = C.c_void_p


def build(targets=("own",), quiet=True):
    """make -C oracle <targets>.  'ref' needs /root/reference (absent on the GPU box: the prebuilt
    oracle/_ref/*.so travel with the repo snapshot instead)."""
    cmd = ["make", "-C", HERE] + list(targets)
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL if quiet else None)


def _load(rel):
    p = os.path.join(HERE, rel)
    if not os.path.exists(p):
        raise FileNotFoundError("%s not built (run `make -C oracle`)" % p)
    return C.CDLL(p)


def have_ref():
    return os.path.exists(os.path.join(HERE, "_ref", "libref_tsdf.so"))


def _p(a):
    return a.ctypes.data_as(_vp)


class OracleVolume:
    """oracle/tsdf_oracle.c behind the same method names as the reference's TSDFVolume."""
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = _load("_build/libtsdf_oracle.so")
            L.oracle_volume_create.restype = _vp
            L.oracle_volume_create.argtypes = [C.c_int, C.c_int, _vp]
            L.oracle_volume_destroy.argtypes = [_vp]
            L.oracle_scale_depth.argtypes = [_vp, _vp, _vp]
            L.oracle_integrate.argtypes = [_vp, _vp, _vp, _vp]
            L.oracle_reproject.argtypes = [_vp, _vp, _vp, C.c_int, C.c_float, _vp, _vp]
            L.oracle_reproject_matrix.argtypes = [_vp, _vp, _vp, _vp]
            L.oracle_compose.argtypes = [_vp, _vp, _vp]
            L.oracle_mat4_inverse.argtypes = [_vp, _vp]
            L.oracle_unit_count.argtypes = [_vp]
            L.oracle_unit_keys.argtypes = [_vp, _vp]
            L.oracle_read_unit.argtypes = [_vp, C.c_int, _vp, _vp]
            L.oracle_sum_weight.restype = C.c_double
            L.oracle_sum_weight.argtypes = [_vp]
            L.oracle_extract_world.restype = C.c_long
            L.oracle_extract_world.argtypes = [_vp, _vp]
            cls._lib = L
        return cls._lib

    def __init__(self, cols=640, rows=480, camera=None):
        L = self.lib()
        self.cols, self.rows = cols, rows
        cam = None if camera is None else np.ascontiguousarray(camera, np.float32)
        self._h = _vp(L.oracle_volume_create(cols, rows, None if cam is None else _p(cam)))

    def close(self):
        if self._h:
            self.lib().oracle_volume_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def ScaleDepth(self, depth):
        d = np.ascontiguousarray(depth, np.uint16).reshape(-1)
        out = np.empty(d.size, np.float32)
        self.lib().oracle_scale_depth(self._h, _p(d), _p(out))
        return out

    def Integrate(self, depth, T, scaled=None):
        d = np.ascontiguousarray(depth, np.uint16).reshape(-1)
        s = self.ScaleDepth(d) if scaled is None else np.ascontiguousarray(scaled, np.float32)
        Tm = np.ascontiguousarray(T, np.float64).reshape(16)
        self.lib().oracle_integrate(self._h, _p(d), _p(s), _p(Tm))

    def Reproject(self, depth, ctr, resolution, length, seg, madj):
        d = np.array(depth, np.uint16).reshape(-1)
        g = np.ascontiguousarray(ctr, np.float32).reshape(-1)
        s = np.ascontiguousarray(seg, np.float64).reshape(16)
        m = np.ascontiguousarray(madj, np.float64).reshape(16)
        self.lib().oracle_reproject(self._h, _p(d), _p(g), int(resolution), C.c_float(length), _p(s), _p(m))
        return d

    @classmethod
    def reproject_matrix(cls, traj_f, traj_0, seg_0):
        a = [np.ascontiguousarray(x, np.float64).reshape(16) for x in (traj_f, traj_0, seg_0)]
        out = np.empty(16, np.float64)
        cls.lib().oracle_reproject_matrix(_p(a[0]), _p(a[1]), _p(a[2]), _p(out))
        return out.reshape(4, 4)

    @classmethod
    def compose(cls, pose, seg):
        a = [np.ascontiguousarray(x, np.float64).reshape(16) for x in (pose, seg)]
        out = np.empty(16, np.float64)
        cls.lib().oracle_compose(_p(a[0]), _p(a[1]), _p(out))
        return out.reshape(4, 4)

    @classmethod
    def inverse(cls, T):
        a = np.ascontiguousarray(T, np.float64).reshape(16)
        out = np.empty(16, np.float64)
        cls.lib().oracle_mat4_inverse(_p(a), _p(out))
        return out.reshape(4, 4)

    def unit_keys(self):
        n = self.lib().oracle_unit_count(self._h)
        k = np.empty(n, np.int32)
        if n:
            self.lib().oracle_unit_keys(self._h, _p(k))
        return k

    def read_unit(self, key):
        sdf = np.empty(UNIT_VOX, np.float32)
        w = np.empty(UNIT_VOX, np.float32)
        if self.lib().oracle_read_unit(self._h, int(key), _p(sdf), _p(w)) != 0:
            raise KeyError(key)
        return sdf, w

    def sum_weight(self):
        return float(self.lib().oracle_sum_weight(self._h))

    def extract_world(self):
        n = self.lib().oracle_extract_world(self._h, None)
        out = np.empty((n, 4), np.float32)
        if n:
            self.lib().oracle_extract_world(self._h, _p(out))
        return out


class RefApp:
    """The reference's CIntegrateApp driven through oracle/ref_driver.cpp (640x480 only, like the reference)."""
    _libs = {}

    @classmethod
    def lib(cls, uncapped=False):
        name = "_ref/libref_tsdf_uncapped.so" if uncapped else "_ref/libref_tsdf.so"
        if name not in cls._libs:
            L = _load(name)
            L.ref_app_create.restype = _vp
            L.ref_app_destroy.argtypes = [_vp]
            L.ref_app_init.argtypes = [_vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int,
                                       C.c_double, C.c_int]
            L.ref_app_set_window.argtypes = [_vp, C.c_int, C.c_int]
            L.ref_app_ctr_num.argtypes = [_vp]
            L.ref_app_execute.argtypes = [_vp, C.c_int, _vp, _vp]
            L.ref_scale_depth.argtypes = [_vp, _vp, _vp]
            L.ref_integrate.argtypes = [_vp, _vp, _vp]
            L.ref_set_camera.argtypes = [_vp, _vp]
            L.ref_unit_count.argtypes = [_vp]
            L.ref_unit_keys.argtypes = [_vp, _vp]
            L.ref_read_unit.argtypes = [_vp, C.c_int, _vp, _vp]
            L.ref_save_world.argtypes = [_vp, C.c_char_p]
            cls._libs[name] = L
        return cls._libs[name]

    def __init__(self, uncapped=False):
        os.environ.setdefault("ER_ORACLE_QUIET", "1")
        self.L = self.lib(uncapped)
        self._h = _vp(self.L.ref_app_create())

    def close(self):
        if self._h:
            self.L.ref_app_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def init(self, ref_traj="", pose_traj="", seg_traj="", ctr="", camera="", num=0, resolution=8, length=3.0, interval=50):
        e = lambda s: (s or "").encode()
        return self.L.ref_app_init(self._h, e(ref_traj), e(pose_traj), e(seg_traj), e(ctr), e(camera), num, resolution,
                                   float(length), interval)

    def set_window(self, start_from, end_at):
        self.L.ref_app_set_window(self._h, start_from, end_at)

    def set_camera(self, cam6):
        c = np.ascontiguousarray(cam6, np.float32)
        self.L.ref_set_camera(self._h, _p(c))

    def execute(self, frame_id, depth, want_scaled=False):
        d = np.array(depth, np.uint16).reshape(-1)
        s = np.empty(d.size, np.float32) if want_scaled else None
        ex = self.L.ref_app_execute(self._h, frame_id, _p(d), None if s is None else _p(s))
        return ex, d, s

    def ScaleDepth(self, depth):
        d = np.ascontiguousarray(depth, np.uint16).reshape(-1)
        out = np.empty(d.size, np.float32)
        self.L.ref_scale_depth(self._h, _p(d), _p(out))
        return out

    def Integrate(self, depth, T):
        d = np.ascontiguousarray(depth, np.uint16).reshape(-1)
        Tm = np.ascontiguousarray(T, np.float64).reshape(16)
        self.L.ref_integrate(self._h, _p(d), _p(Tm))

    def unit_keys(self):
        n = self.L.ref_unit_count(self._h)
        k = np.empty(n, np.int32)
        if n:
            self.L.ref_unit_keys(self._h, _p(k))
        return k

    def read_unit(self, key):
        sdf = np.empty(UNIT_VOX, np.float32)
        w = np.empty(UNIT_VOX, np.float32)
        if self.L.ref_read_unit(self._h, int(key), _p(sdf), _p(w)) != 0:
            raise KeyError(key)
        return sdf, w

    def save_world(self, filename):
        self.L.ref_save_world(self._h, filename.encode())


class IcpOracle:
    """oracle/icp_oracle.cpp: one fragment (xyz + normals) with its CPU search grid.  Methods take the
    TARGET as first argument and use self as the SOURCE, like pcd1 (source) vs pcd0 (target) in
    CorresApp.cpp:246-247."""
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = _load("_build/libicp_oracle.so")
            L.icp_cloud_create.restype = _vp
            L.icp_cloud_create.argtypes = [_vp, _vp, C.c_int, C.c_float]
            L.icp_cloud_destroy.argtypes = [_vp]
            L.icp_nn_pass.argtypes = [_vp, _vp, _vp, C.c_double, _vp, _vp]
            L.icp_count_inliers.argtypes = [_vp, _vp, _vp, C.c_double]
            L.icp_align.argtypes = [_vp, _vp, _vp, C.c_double, C.c_int, C.c_double, C.c_int, _vp, _vp, _vp, _vp]
            L.icp_find_correspondence.argtypes = [_vp, _vp, _vp, C.c_double, C.c_double, _vp, C.c_int, _vp, _vp]
            L.icp_ransac_fitness.argtypes = [_vp, _vp, _vp, C.c_float, _vp, _vp]
            L.icp_ransac_inliers.argtypes = [_vp, _vp, _vp, C.c_float, _vp, _vp, _vp]
            cls._lib = L
        return cls._lib

    def __init__(self, xyz, normals, grid_cell=0.03):
        L = self.lib()
        self.xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        self.nrm = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        self.n = self.xyz.shape[0]
        self._h = _vp(L.icp_cloud_create(_p(self.xyz), _p(self.nrm), self.n, C.c_float(grid_cell)))

    def close(self):
        if self._h:
            self.lib().icp_cloud_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def nn_pass(self, tgt, T, max_dist):
        Tm = np.ascontiguousarray(T, np.float64).reshape(16)
        idx = np.empty(self.n, np.int32)
        sqd = np.empty(self.n, np.float32)
        self.lib().icp_nn_pass(self._h, tgt._h, _p(Tm), float(max_dist), _p(idx), _p(sqd))
        return idx, sqd

    def count_inliers(self, tgt, T, max_dist):
        Tm = np.ascontiguousarray(T, np.float64).reshape(16)
        return int(self.lib().icp_count_inliers(self._h, tgt._h, _p(Tm), float(max_dist)))

    def align(self, tgt, guess, max_dist=0.03, max_iter=20, eps=1e-6, stop_rule=0, want_fitness=False):
        g = np.ascontiguousarray(guess, np.float32).reshape(16)
        out = np.empty(16, np.float32)
        it, cv, fit = C.c_int(0), C.c_int(0), C.c_double(0)
        self.lib().icp_align(self._h, tgt._h, _p(g), float(max_dist), int(max_iter), float(eps), int(stop_rule), _p(out),
                             C.byref(it), C.byref(cv), C.byref(fit) if want_fitness else None)
        return out.reshape(4, 4), it.value, bool(cv.value), (fit.value if want_fitness else None)

    def ransac_fitness(self, tgt, M, corr_dist_threshold):
        """RansacCurvature::getFitness for one hypothesis: (inliers, float32 fitness as the reference sums it, float64 sum)."""
        Mm = np.ascontiguousarray(M, np.float32).reshape(16)
        f, s = C.c_float(0), C.c_double(0)
        cnt = self.lib().icp_ransac_fitness(self._h, tgt._h, _p(Mm), C.c_float(corr_dist_threshold), C.byref(f), C.byref(s))
        return int(cnt), float(f.value), float(s.value)

    def ransac_inliers(self, tgt, M, corr_dist_threshold):
        """getFitness's inlier lists + getInformation: (inliers, inliers_target, information_source, information_target)."""
        Mm = np.ascontiguousarray(M, np.float32).reshape(16)
        pairs = np.empty((max(self.n, 1), 2), np.int32)
        i_s, i_t = np.zeros(36), np.zeros(36)
        m = self.lib().icp_ransac_inliers(self._h, tgt._h, _p(Mm), C.c_float(corr_dist_threshold), _p(pairs), _p(i_s), _p(i_t))
        return pairs[:m, 0].copy(), pairs[:m, 1].copy(), i_s.reshape(6, 6), i_t.reshape(6, 6)

    def find_correspondence(self, tgt, T, dist, normal_cos=0.8660, want_info=False):
        Tm = np.ascontiguousarray(T, np.float64).reshape(16)
        pairs = np.empty((max(self.n, 1), 2), np.int32)
        m = C.c_int(0)
        info = np.zeros(36, np.float64) if want_info else None
        self.lib().icp_find_correspondence(self._h, tgt._h, _p(Tm), float(dist), float(normal_cos), _p(pairs), self.n,
                                           C.byref(m), _p(info) if want_info else None)
        return pairs[:m.value].copy(), (info.reshape(6, 6) if want_info else None)


class FoptOracle:
    """oracle/fopt_oracle.cpp: FragmentOptimizer's point state and Hessian assembly (rigid / SLAC), SURVEY.md 8f-2."""
    _lib = None

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = _load("_build/libfopt_oracle.so")
            L.fopt_create.restype = _vp
            L.fopt_create.argtypes = [C.c_int, C.c_int, C.c_float]
            L.fopt_destroy.argtypes = [_vp]
            L.fopt_set_cloud.argtypes = [_vp, C.c_int, _vp, _vp, C.c_int]
            L.fopt_cloud_size.argtypes = [_vp, C.c_int]
            L.fopt_get_points.argtypes = [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp]
            L.fopt_update_pose.argtypes = [_vp, C.c_int, _vp]
            L.fopt_update_point_pn.argtypes = [_vp, C.c_int, _vp]
            L.fopt_clear_pairs.argtypes = [_vp]
            L.fopt_add_pair.argtypes = [_vp, C.c_int, C.c_int, _vp, C.c_int]
            L.fopt_assemble_rigid.argtypes = [_vp, _vp, _vp, _vp]
            L.fopt_assemble_slac.argtypes = [_vp, _vp, _vp, _vp, _vp]
            L.fopt_rigid_bucket.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]
            L.fopt_slac_bucket.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]
            L.fopt_update_normals.argtypes = [_vp, C.c_int, _vp]
            L.fopt_assemble_nonrigid.restype = C.c_long
            L.fopt_assemble_nonrigid.argtypes = [_vp, C.c_double, _vp, _vp]
            L.fopt_nonrigid_bucket.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _vp, _vp, _vp, _vp]
            cls._lib = L
        return cls._lib

    def __init__(self, num, resolution=8, length=3.0):
        self.num, self.resolution, self.length = num, resolution, float(length)
        self.nper = (resolution + 1) ** 3 * 3
        self._h = _vp(self.lib().fopt_create(num, resolution, C.c_float(length)))

    def close(self):
        if self._h:
            self.lib().fopt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_cloud(self, frag, xyz, nrm):
        x = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        n = np.ascontiguousarray(nrm, np.float32).reshape(-1, 3)
        return int(self.lib().fopt_set_cloud(self._h, frag, _p(x), _p(n), x.shape[0]))

    def points(self, frag):
        m = int(self.lib().fopt_cloud_size(self._h, frag))
        idx0, val, nval = np.zeros(m, np.int32), np.zeros((m, 8), np.float32), np.zeros((m, 8), np.float32)
        p, n = np.zeros((m, 3), np.float32), np.zeros((m, 3), np.float32)
        self.lib().fopt_get_points(self._h, frag, _p(idx0), _p(val), _p(nval), _p(p), _p(n))
        return dict(idx0=idx0, val=val, nval=nval, p=p, n=n)

    def update_pose(self, frag, M):
        Mm = np.ascontiguousarray(M, np.float32).reshape(16)
        self.lib().fopt_update_pose(self._h, frag, _p(Mm))

    def update_point_pn(self, frag, ctr_slice):
        c = np.ascontiguousarray(ctr_slice, np.float64).reshape(-1)
        assert c.size == self.nper
        self.lib().fopt_update_point_pn(self._h, frag, _p(c))

    def set_pairs(self, pairs):
        """pairs: list of (i, j, int32 [m,2] rows (index in i, index in j))."""
        self.lib().fopt_clear_pairs(self._h)
        for i, j, pr in pairs:
            a = np.ascontiguousarray(pr, np.int32).reshape(-1, 2)
            self.lib().fopt_add_pair(self._h, int(i), int(j), _p(a), a.shape[0])

    def assemble_rigid(self):
        N = 6 * self.num
        JJ, Jb, sc = np.zeros((N, N)), np.zeros(N), C.c_double(0)
        self.lib().fopt_assemble_rigid(self._h, _p(JJ), _p(Jb), C.byref(sc))
        return JJ, Jb, sc.value

    def assemble_slac(self, pose_rot_t):
        N = 6 * self.num + self.nper
        R = np.ascontiguousarray(pose_rot_t, np.float64).reshape(self.num, 9)
        JJ, Jb, sc = np.zeros((N, N)), np.zeros(N), C.c_double(0)
        self.lib().fopt_assemble_slac(self._h, _p(R), _p(JJ), _p(Jb), C.byref(sc))
        return JJ, Jb, sc.value

    def update_normals(self, frag, ctr_slice):
        c = np.ascontiguousarray(ctr_slice, np.float64).reshape(-1)
        assert c.size == self.nper
        self.lib().fopt_update_normals(self._h, frag, _p(c))

    def assemble_nonrigid(self, weight):
        """OptimizeNonrigid's data term as merged triplets: (rows, cols, vals), global indices fragment * nper + lattice index."""
        n = int(self.lib().fopt_assemble_nonrigid(self._h, float(weight), None, None))
        keys, vals = np.zeros(n, np.int64), np.zeros(n, np.float64)
        self.lib().fopt_assemble_nonrigid(self._h, float(weight), _p(keys), _p(vals))
        M = self.nper * self.num
        return keys // M, keys % M, vals

    def nonrigid_bucket(self, i, ii, j, jj, weight):
        i1, v1, i2, v2 = np.zeros(24, np.int32), np.zeros(24), np.zeros(24, np.int32), np.zeros(24)
        self.lib().fopt_nonrigid_bucket(self._h, i, ii, j, jj, float(weight), _p(i1), _p(v1), _p(i2), _p(v2))
        return i1, v1, i2, v2

    def rigid_bucket(self, i, ii, j, jj):
        val, b = np.zeros(12), C.c_double(0)
        self.lib().fopt_rigid_bucket(self._h, i, ii, j, jj, _p(val), C.byref(b))
        return val, b.value

    def slac_bucket(self, i, ii, j, jj, pose_rot_t):
        R = np.ascontiguousarray(pose_rot_t, np.float64).reshape(self.num, 9)
        idx, val, b = np.zeros(60, np.int32), np.zeros(60), C.c_double(0)
        self.lib().fopt_slac_bucket(self._h, i, ii, j, jj, _p(R), _p(idx), _p(val), C.byref(b))
        return idx, val, b.value


class RefFopt:
    """oracle/_ref/libref_fopt.so: the reference's own PointCloud.{h,cpp} compiled in place (point state only).
    Only available where /root/reference was present at build time."""
    _lib = None

    @classmethod
    def available(cls):
        return os.path.exists(os.path.join(HERE, "_ref", "libref_fopt.so"))

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = _load("_ref/libref_fopt.so")
            L.rfopt_cloud_create.restype = _vp
            L.rfopt_cloud_create.argtypes = [C.c_int, C.c_int, C.c_float]
            L.rfopt_cloud_destroy.argtypes = [_vp]
            L.rfopt_cloud_load.argtypes = [_vp, _vp, _vp, C.c_int]
            L.rfopt_get_points.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp]
            L.rfopt_update_pose.argtypes = [_vp, _vp]
            L.rfopt_update_point_pn.argtypes = [_vp, _vp, C.c_int]
            L.rfopt_update_normals.argtypes = [_vp, _vp, C.c_int]
            cls._lib = L
        return cls._lib

    def __init__(self, num, resolution=8, length=3.0):
        self.num, self.resolution = num, resolution
        self.nper = (resolution + 1) ** 3 * 3
        self.clouds = [_vp(self.lib().rfopt_cloud_create(i, resolution, C.c_float(length))) for i in range(num)]
        self.sizes = [0] * num

    def close(self):
        for c in self.clouds:
            self.lib().rfopt_cloud_destroy(c)
        self.clouds = []

    def set_cloud(self, frag, xyz, nrm):
        x = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        n = np.ascontiguousarray(nrm, np.float32).reshape(-1, 3)
        r = int(self.lib().rfopt_cloud_load(self.clouds[frag], _p(x), _p(n), x.shape[0]))
        self.sizes[frag] = x.shape[0] if r < 0 else r + 1
        return r

    def points(self, frag):
        m = self.sizes[frag]
        idx0, val, nval = np.zeros(m, np.int32), np.zeros((m, 8), np.float32), np.zeros((m, 8), np.float32)
        p, n = np.zeros((m, 3), np.float32), np.zeros((m, 3), np.float32)
        self.lib().rfopt_get_points(self.clouds[frag], _p(idx0), _p(val), _p(nval), _p(p), _p(n))
        return dict(idx0=idx0, val=val, nval=nval, p=p, n=n)

    def update_pose(self, frag, M):
        Mm = np.ascontiguousarray(M, np.float32).reshape(16)
        self.lib().rfopt_update_pose(self.clouds[frag], _p(Mm))

    def update_point_pn(self, frag, ctr_full):
        c = np.ascontiguousarray(ctr_full, np.float64).reshape(-1)
        self.lib().rfopt_update_point_pn(self.clouds[frag], _p(c), c.size)

    def update_normals(self, frag, ctr_full):
        c = np.ascontiguousarray(ctr_full, np.float64).reshape(-1)
        self.lib().rfopt_update_normals(self.clouds[frag], _p(c), c.size)


class RefCorres:
    """oracle/_ref/libref_corres*.so: the reference's own CCorresApp (BuildCorrespondence/CorresApp.{h,cpp} compiled in place,
    unmodified, against oracle/stub_corres) on in-memory clouds.  Registration's pre-check / accept rule, FindCorrespondence, the
    ratio test and the information matrix run as REFERENCE code; the PCL calls inside them are the stub's (exact kd-tree, PCL 1.7
    transforms and ICP restated on the reference's vendored Eigen).  Only where /root/reference was present at build time."""
    _libs = {}
    REF_BIN = os.path.join(HERE, "_ref", "BuildCorrespondence_ref")

    @classmethod
    def available(cls):
        return os.path.exists(os.path.join(HERE, "_ref", "libref_corres.so"))

    @classmethod
    def lib(cls, uncapped=False):
        if uncapped not in cls._libs:
            L = _load("_ref/libref_corres_uncapped.so" if uncapped else "_ref/libref_corres.so")
            L.rcorres_create.restype = _vp
            L.rcorres_create.argtypes = [C.c_char_p]
            L.rcorres_destroy.argtypes = [_vp]
            L.rcorres_set_params.argtypes = [_vp, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
            L.rcorres_add_cloud.argtypes = [_vp, _vp, _vp, C.c_int]
            L.rcorres_add_pair.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp]
            L.rcorres_blacklist.argtypes = [_vp, C.c_int]
            L.rcorres_registration.argtypes = [_vp]
            L.rcorres_find_correspondence.argtypes = [_vp]
            L.rcorres_num_pairs.argtypes = [_vp]
            L.rcorres_get_pair.argtypes = [_vp, C.c_int, _vp, _vp, _vp]
            L.rcorres_overlap_ratio.restype = C.c_double
            L.rcorres_overlap_ratio.argtypes = [_vp, C.c_double, _vp]
            L.rcorres_icp.argtypes = [_vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, C.c_double, C.c_int, C.c_double, _vp, _vp, _vp, _vp]
            cls._libs[uncapped] = L
        return cls._libs[uncapped]

    def __init__(self, out_dir=None, uncapped=False, reg_dist=0.03, dist_thresh=None, reg_ratio=0.25, reg_num=40000,
                 output_information=True):
        """out_dir (with trailing slash): where FindCorrespondence writes corres_<i>_<j>.txt; None = save_corres_ off."""
        self.L = self.lib(uncapped)
        self._h = _vp(self.L.rcorres_create(out_dir.encode() if out_dir else None))
        self.L.rcorres_set_params(self._h, float(reg_dist), float(reg_dist / 2.0 if dist_thresh is None else dist_thresh),
                                  float(reg_ratio), int(reg_num), int(bool(output_information)))
        self.output_information = bool(output_information)

    def close(self):
        if self._h:
            self.L.rcorres_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_cloud(self, xyz, nrm):
        x = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        n = np.ascontiguousarray(nrm, np.float32).reshape(-1, 3)
        return int(self.L.rcorres_add_cloud(self._h, _p(x), _p(n), x.shape[0]))

    def add_pair(self, id1, id2, frame, T):
        Tm = np.ascontiguousarray(T, np.float64).reshape(16)
        self.L.rcorres_add_pair(self._h, int(id1), int(id2), int(frame), _p(Tm))

    def blacklist(self, i):
        self.L.rcorres_blacklist(self._h, int(i))

    def Registration(self):
        self.L.rcorres_registration(self._h)

    def FindCorrespondence(self):
        self.L.rcorres_find_correspondence(self._h)

    def pairs(self):
        """[(id1, id2, frame, T float64 4x4, information 6x6 or None)] = corres_traj_ / corres_info_ as they stand."""
        out = []
        for k in range(int(self.L.rcorres_num_pairs(self._h))):
            ids = np.zeros(3, np.int32)
            T, info = np.zeros(16), np.zeros(36)
            self.L.rcorres_get_pair(self._h, k, _p(ids), _p(T), _p(info) if self.output_information else None)
            out.append((int(ids[0]), int(ids[1]), int(ids[2]), T.reshape(4, 4), info.reshape(6, 6) if self.output_information else None))
        return out

    def overlap_ratio(self, T, length=3.0):
        Tm = np.ascontiguousarray(T, np.float64).reshape(16)
        return float(self.L.rcorres_overlap_ratio(self._h, float(length), _p(Tm)))

    @classmethod
    def icp(cls, src_xyz, src_nrm, tgt_xyz, tgt_nrm, guess, max_dist=0.03, max_iter=20, eps=1e-6, want_fitness=False):
        """The stub's PCL 1.7 ICP (kd-tree + vendored Eigen) as CorresApp.cpp:295-306 configures it: (T float32, iterations, converged, fitness)."""
        L = cls.lib()
        sx, sn = np.ascontiguousarray(src_xyz, np.float32), np.ascontiguousarray(src_nrm, np.float32)
        tx, tn = np.ascontiguousarray(tgt_xyz, np.float32), np.ascontiguousarray(tgt_nrm, np.float32)
        g = np.ascontiguousarray(guess, np.float32).reshape(16)
        out = np.empty(16, np.float32)
        it, cv, fit = C.c_int(0), C.c_int(0), C.c_double(0)
        L.rcorres_icp(_p(sx), _p(sn), sx.shape[0], _p(tx), _p(tn), tx.shape[0], _p(g), float(max_dist), int(max_iter), float(eps),
                      _p(out), C.byref(it), C.byref(cv), C.byref(fit) if want_fitness else None)
        return out.reshape(4, 4), it.value, bool(cv.value), (fit.value if want_fitness else None)


class RefRansac:
    """oracle/_ref/libref_ransac.so: the reference's own GlobalRegistration/RansacCurvature.h (getFitness, getInformation,
    align_redux) included in place behind oracle/ref_ransac_driver.cpp.  Only where /root/reference was present at build time."""
    _lib = None

    @classmethod
    def available(cls):
        return os.path.exists(os.path.join(HERE, "_ref", "libref_ransac.so"))

    @classmethod
    def lib(cls):
        if cls._lib is None:
            L = _load("_ref/libref_ransac.so")
            L.rransac_create.restype = _vp
            L.rransac_create.argtypes = [_vp, _vp, C.c_int, _vp, _vp, C.c_int, C.c_double, C.c_float, C.c_int]
            L.rransac_destroy.argtypes = [_vp]
            L.rransac_fitness.argtypes = [_vp, _vp, _vp, _vp, _vp]
            L.rransac_align_redux.argtypes = [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]
            cls._lib = L
        return cls._lib

    def __init__(self, src_xyz, src_nrm, tgt_xyz, tgt_nrm, max_corr_dist, inlier_fraction=0.0, inlier_number=1000000):
        sx, sn = np.ascontiguousarray(src_xyz, np.float32), np.ascontiguousarray(src_nrm, np.float32)
        tx, tn = np.ascontiguousarray(tgt_xyz, np.float32), np.ascontiguousarray(tgt_nrm, np.float32)
        self.n = sx.shape[0]
        self._h = _vp(self.lib().rransac_create(_p(sx), _p(sn), self.n, _p(tx), _p(tn), tx.shape[0], float(max_corr_dist),
                                                C.c_float(inlier_fraction), int(inlier_number)))

    def close(self):
        if self._h:
            self.lib().rransac_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fitness(self, M):
        """getFitness with final_transformation_ = M: (inliers, inliers_target, float32 fitness score)."""
        Mm = np.ascontiguousarray(M, np.float32).reshape(16)
        a, b = np.empty(max(self.n, 1), np.int32), np.empty(max(self.n, 1), np.int32)
        f = C.c_float(0)
        m = int(self.lib().rransac_fitness(self._h, _p(Mm), _p(a), _p(b), C.byref(f)))
        return a[:m].copy(), b[:m].copy(), float(f.value)

    def align_redux(self, guess):
        """align_redux( out, guess ) + getInformation(): (converged, inliers_, inliers_target_, information_source_, information_target_)."""
        g = np.ascontiguousarray(guess, np.float32).reshape(16)
        a, b = np.empty(max(self.n, 1), np.int32), np.empty(max(self.n, 1), np.int32)
        i_s, i_t = np.zeros(36), np.zeros(36)
        m = C.c_int(0)
        conv = int(self.lib().rransac_align_redux(self._h, _p(g), self.n, C.byref(m), _p(a), _p(b), _p(i_s), _p(i_t)))
        return bool(conv), a[:m.value].copy(), b[:m.value].copy(), i_s.reshape(6, 6), i_t.reshape(6, 6)
