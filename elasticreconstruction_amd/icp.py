"""Host-side mirror of the reference's BuildCorrespondence program over the C ABI (include/er_hip.h).

  Cloud      <->  pointclouds_[i]  (BuildCorrespondence/CorresApp.h:32) resident in HBM with its search grid
  CorresApp  <->  CCorresApp       (CorresApp.h:12-82, CorresApp.cpp): LoadData / Registration /
                  FindCorrespondence / Finalize / Blacklist / Redux with the reference's member names,
                  defaults and output files (reg_output.log/.info in the CWD, corres_<i>_<j>.txt next to the clouds).

All per-point arithmetic (transform, exact NN, inlier counts, point-to-plane sums, correspondence filter,
information matrix) runs in the HIP kernels of liber_hip.so.
"""
import ctypes as C
import os

import numpy as np

from . import _ffi
from . import formats
from .tsdf import _inverse, mat4_mul


class Cloud:
    """One fragment in HBM: xyz + normals (file order) and the uniform grid used when it is a target."""

    def __init__(self, xyz, normals, grid_cell=0.03, device=0):
        self._lib = _ffi.lib()
        x = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        n = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
        assert x.shape == n.shape
        self.n = x.shape[0]
        self.grid_cell = float(grid_cell)
        h = C.c_void_p()
        _ffi.check(self._lib.er_cloud_create(_ffi.ptr(x), _ffi.ptr(n), self.n, C.c_float(grid_cell), int(device), C.byref(h)),
                   "er_cloud_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.er_cloud_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.n

    @classmethod
    def create_batch(cls, arrays, grid_cell=0.03, device=0):
        """er_cloud_create_batch: [Cloud] for a list of (xyz, normals) in one call.  Arrays that live in page-locked memory
        (_ffi.PinnedArena) are uploaded asynchronously -- the list is then PCIe-bound."""
        lib = _ffi.lib()
        xs = [np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 3) for x, _ in arrays]
        ns = [np.ascontiguousarray(n, dtype=np.float32).reshape(-1, 3) for _, n in arrays]
        assert all(x.shape == n.shape for x, n in zip(xs, ns))
        m = len(xs)
        px = (C.c_void_p * m)(*[x.ctypes.data for x in xs])
        pn = (C.c_void_p * m)(*[n.ctypes.data for n in ns])
        cnt = np.array([x.shape[0] for x in xs], np.int32)
        hs = (C.c_void_p * m)()
        _ffi.check(lib.er_cloud_create_batch(m, px, pn, _ffi.ptr(cnt), C.c_float(grid_cell), int(device), hs), "er_cloud_create_batch")
        out = []
        for k in range(m):
            c = cls.__new__(cls)
            c._lib, c.n, c.grid_cell, c._h = lib, int(cnt[k]), float(grid_cell), C.c_void_p(hs[k])
            out.append(c)
        return out


def count_inliers(src, tgt, T, max_dist):
    """Registration pre-check (CorresApp.cpp:249-264)."""
    Tm = np.ascontiguousarray(T, np.float64).reshape(16)
    cnt = C.c_int(0)
    _ffi.check(src._lib.er_icp_count_inliers(src._h, tgt._h, _ffi.ptr(Tm), float(max_dist), C.byref(cnt)), "er_icp_count_inliers")
    return cnt.value


def icp_align(src, tgt, guess, max_dist=0.03, max_iter=20, eps=1e-6, stop_rule=0, want_fitness=False):
    """icp.align as configured at CorresApp.cpp:295-306.  Returns (float32 4x4, iterations, converged, fitness)."""
    g = np.ascontiguousarray(guess, np.float32).reshape(16)
    out = np.empty(16, np.float32)
    it, cv, fit = C.c_int(0), C.c_int(0), C.c_double(0)
    _ffi.check(src._lib.er_icp_align(src._h, tgt._h, _ffi.ptr(g), float(max_dist), int(max_iter), float(eps), int(stop_rule),
                                     _ffi.ptr(out), C.byref(it), C.byref(cv), C.byref(fit) if want_fitness else None),
               "er_icp_align")
    return out.reshape(4, 4), it.value, bool(cv.value), (fit.value if want_fitness else None)


def find_correspondence(src, tgt, T, dist, normal_cos=0.8660, want_info=False):
    """FindCorrespondence (CorresApp.cpp:144-161,186-208).  Returns (pairs int32 [m,2] = (tgt idx, src idx), info 6x6 or None)."""
    Tm = np.ascontiguousarray(T, np.float64).reshape(16)
    pairs = np.empty((max(src.n, 1), 2), np.int32)
    m = C.c_int(0)
    info = np.zeros(36, np.float64) if want_info else None
    _ffi.check(src._lib.er_find_correspondence(src._h, tgt._h, _ffi.ptr(Tm), float(dist), float(normal_cos), _ffi.ptr(pairs),
                                               src.n, C.byref(m), _ffi.ptr(info) if want_info else None), "er_find_correspondence")
    return pairs[:m.value].copy(), (info.reshape(6, 6) if want_info else None)


def _handles(clouds):
    return (C.c_void_p * len(clouds))(*[c._h for c in clouds])


def count_inliers_batch(srcs, tgts, Ts, max_dist):
    """The pre-check of every pair of one Registration loop in one call (pairs pipelined over several streams)."""
    n = len(srcs)
    if n == 0:
        return np.zeros(0, np.int32)
    Tm = np.ascontiguousarray(Ts, np.float64).reshape(n, 16)
    out = np.zeros(n, np.int32)
    _ffi.check(srcs[0]._lib.er_icp_count_inliers_batch(n, _handles(srcs), _handles(tgts), _ffi.ptr(Tm), float(max_dist), _ffi.ptr(out)),
               "er_icp_count_inliers_batch")
    return out


def icp_align_batch(srcs, tgts, guesses, max_dist=0.03, max_iter=20, eps=1e-6, stop_rule=0, want_fitness=False):
    """icp.align for a list of pairs.  Returns (float32 [n,4,4], iterations [n], converged [n], fitness [n] or None)."""
    n = len(srcs)
    g = np.ascontiguousarray(guesses, np.float32).reshape(n, 16)
    out = np.empty((n, 16), np.float32)
    it, cv = np.zeros(n, np.int32), np.zeros(n, np.int32)
    fit = np.zeros(n, np.float64) if want_fitness else None
    if n:
        _ffi.check(srcs[0]._lib.er_icp_align_batch(n, _handles(srcs), _handles(tgts), _ffi.ptr(g), float(max_dist), int(max_iter), float(eps),
                                                   int(stop_rule), _ffi.ptr(out), _ffi.ptr(it), _ffi.ptr(cv),
                                                   _ffi.ptr(fit) if want_fitness else None), "er_icp_align_batch")
    return out.reshape(n, 4, 4), it, cv.astype(bool), fit


def ransac_fitness_batch(src, tgt, Ms, corr_dist_threshold):
    """RansacCurvature::getFitness (GlobalRegistration/RansacCurvature.h:661-704) for a stack of float32 4x4 hypotheses of one
    pair.  Returns (inlier counts int32 [n], fitness float64 [n])."""
    M = np.ascontiguousarray(Ms, np.float32).reshape(-1, 16)
    n = M.shape[0]
    cnt, fit = np.zeros(n, np.int32), np.zeros(n, np.float64)
    _ffi.check(src._lib.er_ransac_fitness_batch(src._h, tgt._h, n, _ffi.ptr(M), C.c_float(corr_dist_threshold), _ffi.ptr(cnt), _ffi.ptr(fit)),
               "er_ransac_fitness_batch")
    return cnt, fit


def ransac_inliers(src, tgt, M, corr_dist_threshold):
    """getFitness's inlier lists for one hypothesis + getInformation (GlobalRegistration/RansacCurvature.h:661-733).
    Returns (inliers int32 [m], inliers_target int32 [m], fitness, information_source [6,6], information_target [6,6])."""
    Mf = np.ascontiguousarray(M, np.float32).reshape(16)
    pairs = np.zeros((max(src.n, 1), 2), np.int32)
    m = C.c_int(0)
    fit = C.c_double(0.0)
    info_s, info_t = np.zeros((6, 6)), np.zeros((6, 6))
    _ffi.check(src._lib.er_ransac_inliers(src._h, tgt._h, _ffi.ptr(Mf), C.c_float(corr_dist_threshold), _ffi.ptr(pairs), src.n, C.byref(m),
                                          C.byref(fit), _ffi.ptr(info_s), _ffi.ptr(info_t)), "er_ransac_inliers")
    pairs = pairs[:m.value]
    return pairs[:, 1].copy(), pairs[:, 0].copy(), fit.value, info_s, info_t      # pairs are (target, source) like corres_*.txt


_arena = None


def find_correspondence_batch(srcs, tgts, Ts, dist, normal_cos=0.8660, want_info=False, copy=True):
    """FindCorrespondence for a list of pairs.  Returns ([pairs int32 [m_i,2]], info [n,6,6] or None).
    The lists are written into a process-wide page-locked arena (no staging copy on the way from the GPU);
    copy=False returns views into it, valid until the next find_correspondence_batch call."""
    global _arena
    n = len(srcs)
    if n == 0:
        return [], (np.zeros((0, 6, 6)) if want_info else None)
    if _arena is None:
        _arena = _ffi.PinnedArena()
    Tm = np.ascontiguousarray(Ts, np.float64).reshape(n, 16)
    # ONE block of the arena for all lists (a numpy view per pair costs ~10 us; 50 of them were 7 % of a 50-pair pass): pair i's
    # buffer starts at a 4 KiB boundary inside it, the pointers are plain address arithmetic
    cap = np.array([s.n for s in srcs], np.int32)
    ints = ((np.maximum(cap, 1).astype(np.int64) * 2 + 1023) // 1024) * 1024
    offs = np.concatenate([[0], np.cumsum(ints)[:-1]])
    _arena.reset(int(ints.sum()) * 4 + 8192)
    big = _arena.take((int(ints.sum()),), np.int32)
    base = big.ctypes.data
    ptrs = (C.c_void_p * n)(*(base + 4 * offs).tolist())
    m = np.zeros(n, np.int32)
    info = np.zeros((n, 36), np.float64) if want_info else None
    _ffi.check(srcs[0]._lib.er_find_correspondence_batch(n, _handles(srcs), _handles(tgts), _ffi.ptr(Tm), float(dist), float(normal_cos),
                                                         ptrs, _ffi.ptr(cap), _ffi.ptr(m), _ffi.ptr(info) if want_info else None),
               "er_find_correspondence_batch")
    lists = [big[o:o + 2 * k].reshape(k, 2) for o, k in zip(offs.tolist(), m.tolist())]
    return [(l.copy() if copy else l) for l in lists], (info.reshape(n, 6, 6) if want_info else None)


def registration_batch(srcs, tgts, Ts, reg_dist=0.03, reg_num=40000, reg_ratio=0.25, max_iter=20, eps=1e-6, stop_rule=0, corr_dist=0.015,
                       normal_cos=0.8660, want_info=False, copy=True):
    """er_registration_batch: CCorresApp::Registration + FindCorrespondence over a pair list in one call (CorresApp.cpp:212-319, 112-210).
    Returns dict(counts, accepted, T (float32 [n,4,4]), iterations, converged, lists ([int32 [m_i,2]], empty for rejected pairs), info or None).
    copy=False: the lists are views into the process-wide page-locked arena, valid until the next *_batch call that returns lists."""
    global _arena
    n = len(srcs)
    if n == 0:
        return dict(counts=np.zeros(0, np.int32), accepted=np.zeros(0, bool), T=np.zeros((0, 4, 4), np.float32), iterations=np.zeros(0, np.int32),
                    converged=np.zeros(0, bool), lists=[], info=np.zeros((0, 6, 6)) if want_info else None)
    if _arena is None:
        _arena = _ffi.PinnedArena()
    Tm = np.ascontiguousarray(Ts, np.float64).reshape(n, 16)
    cap = np.array([s.n for s in srcs], np.int32)
    ints = ((np.maximum(cap, 1).astype(np.int64) * 2 + 1023) // 1024) * 1024
    offs = np.concatenate([[0], np.cumsum(ints)[:-1]])
    _arena.reset(int(ints.sum()) * 4 + 8192)
    big = _arena.take((int(ints.sum()),), np.int32)
    ptrs = (C.c_void_p * n)(*(big.ctypes.data + 4 * offs).tolist())
    counts, acc, its, cv, m = (np.zeros(n, np.int32) for _ in range(5))
    F = np.zeros((n, 16), np.float32)
    info = np.zeros((n, 36), np.float64) if want_info else None
    _ffi.check(srcs[0]._lib.er_registration_batch(n, _handles(srcs), _handles(tgts), _ffi.ptr(Tm), float(reg_dist), int(reg_num), float(reg_ratio),
                                                  int(max_iter), float(eps), int(stop_rule), float(corr_dist), float(normal_cos), _ffi.ptr(counts),
                                                  _ffi.ptr(acc), _ffi.ptr(F), _ffi.ptr(its), _ffi.ptr(cv), ptrs, _ffi.ptr(cap), _ffi.ptr(m),
                                                  _ffi.ptr(info) if want_info else None), "er_registration_batch")
    lists = [big[o:o + 2 * k].reshape(k, 2) for o, k in zip(offs.tolist(), m.tolist())]
    return dict(counts=counts, accepted=acc.astype(bool), T=F.reshape(n, 4, 4), iterations=its, converged=cv.astype(bool),
                lists=[(l.copy() if copy else l) for l in lists], info=info.reshape(n, 6, 6) if want_info else None)


class DeviceLists:
    """Correspondence lists of a pair list that STAY IN HBM (round 5): one er_device_alloc block cut into one slot per pair (room for |source|
    rows each), handed to er_registration_batch / er_find_correspondence_batch as their list buffers -- a buffer may be device memory, the
    copy is then device-to-device -- and on to er_fopt_set_correspondences_dev.  counts[k] = rows of pair k after the call."""

    def __init__(self, srcs, device=0):
        self._lib = _ffi.lib()
        self.n = len(srcs)
        self.cap = np.array([s.n for s in srcs], np.int32)
        ints = ((np.maximum(self.cap, 1).astype(np.int64) * 2 + 63) // 64) * 64
        self.offs = np.concatenate([[0], np.cumsum(ints)[:-1]]).astype(np.int64)
        self.total_ints = int(ints.sum())
        self.base = self._lib.er_device_alloc(self.total_ints * 4, int(device))
        if not self.base:
            raise _ffi.ErError("er_device_alloc: " + self._lib.er_last_error().decode())
        self.counts = np.zeros(self.n, np.int32)

    def ptrs(self):
        return (C.c_void_p * max(self.n, 1))(*[self.base + 4 * int(o) for o in self.offs])

    def download(self, k):
        """One list to the host (when a corres_<i>_<j>.txt is wanted after all)."""
        m = int(self.counts[k])
        out = np.empty((m, 2), np.int32)
        if m:
            _ffi.check(self._lib.er_device_copy_d2h(_ffi.ptr(out), C.c_void_p(self.base + 4 * int(self.offs[k])), m * 8), "er_device_copy_d2h")
        return out

    def close(self):
        if getattr(self, "base", None):
            self._lib.er_device_free(C.c_void_p(self.base))
            self.base = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def registration_batch_dev(srcs, tgts, Ts, lists, reg_dist=0.03, reg_num=40000, reg_ratio=0.25, max_iter=20, eps=1e-6, stop_rule=0, corr_dist=0.015,
                           normal_cos=0.8660, want_info=False):
    """er_registration_batch with the correspondence lists left in HBM (`lists`: a DeviceLists over the same sources).  Everything else comes
    back as registration_batch returns it; lists.counts holds the list lengths."""
    n = len(srcs)
    Tm = np.ascontiguousarray(Ts, np.float64).reshape(n, 16)
    counts, acc, its, cv = (np.zeros(n, np.int32) for _ in range(4))
    F = np.zeros((n, 16), np.float32)
    info = np.zeros((n, 36), np.float64) if want_info else None
    _ffi.check(srcs[0]._lib.er_registration_batch(n, _handles(srcs), _handles(tgts), _ffi.ptr(Tm), float(reg_dist), int(reg_num), float(reg_ratio),
                                                  int(max_iter), float(eps), int(stop_rule), float(corr_dist), float(normal_cos), _ffi.ptr(counts),
                                                  _ffi.ptr(acc), _ffi.ptr(F), _ffi.ptr(its), _ffi.ptr(cv), lists.ptrs(), _ffi.ptr(lists.cap), _ffi.ptr(lists.counts),
                                                  _ffi.ptr(info) if want_info else None), "er_registration_batch")
    return dict(counts=counts, accepted=acc.astype(bool), T=F.reshape(n, 4, 4), iterations=its, converged=cv.astype(bool),
                info=info.reshape(n, 6, 6) if want_info else None)


class CorresApp:
    """CCorresApp (CorresApp.h:12-82).  Defaults from the constructor, CorresApp.cpp:8-24."""

    def __init__(self, device=0, verbose=False):
        self.save_xyzn_ = False
        self.save_corres_ = True
        self.dist_thresh_ = 0.015
        self.normal_thresh_ = 0.8660
        self.registration_ = False
        self.output_information_ = False
        self.reg_dist_ = 0.03
        self.reg_ratio_ = 0.25
        self.reg_num_ = 40000
        self.redux_ = False
        self.num_ = 0
        self.length_ = 3.0
        self.interval_ = 50
        self.corres_traj_ = []
        self.corres_info_ = []
        self.blacklist_ = set()
        self.redux_traj_ = []
        self.redux_map_ = {}
        self.pointclouds_ = []
        self.m_pDirName = ""
        self.device = device
        self.verbose = verbose
        self.stop_rule = 0
        self.out_dir = "."           # reg_output.* go to the CWD in the reference (CorresApp.cpp:323,326)
        self.icp_iterations_ = {}
        self.keep_correspondences_ = True   # also keep the lists in memory (tests, small runs)

    # ---- CorresApp.h:64-81 -----------------------------------------------------------------------
    def GetVolumeOverlapRatio(self, trans):
        res = 20
        ul = self.length_ / float(res)
        c = (np.arange(res) + 0.5) * ul
        i, j, k = np.meshgrid(c, c, c, indexing="ij")
        T = np.asarray(trans, np.float64)
        p = [((T[r, 0] * i + T[r, 1] * j) + T[r, 2] * k) + T[r, 3] * 1.0 for r in range(3)]
        inside = np.ones_like(i, dtype=bool)
        for a in range(3):
            inside &= (p[a] >= 0) & (p[a] <= self.length_)
        return float(inside.sum()) / res / res / res

    def InitialPairs(self, filename, num):
        """The --traj/--num branch of LoadData (CorresApp.cpp:40-69): fragment poses from every interval_-th camera pose,
        consecutive fragments always paired, the others when their cubes overlap by more than 30 %.  Host-only."""
        temp = formats.load_log(filename)
        self.corres_traj_ = []
        self.num_ = num
        base = np.eye(4)
        base[0, 3] = self.length_ / 2.0
        base[1, 3] = self.length_ / 2.0
        base[2, 3] = -0.3
        baseinv = _inverse(base)
        leftbase = mat4_mul(base, _inverse(temp[0].T))
        ipose = [mat4_mul(mat4_mul(leftbase, temp[i * self.interval_].T), baseinv) for i in range(num)]
        for i in range(num - 1):
            self.corres_traj_.append(formats.FramedTransformation(i, i + 1, num, mat4_mul(_inverse(ipose[i]), ipose[i + 1])))
            for j in range(i + 2, num):
                trans = mat4_mul(_inverse(ipose[i]), ipose[j])
                if self.GetVolumeOverlapRatio(trans) > 0.3:
                    self.corres_traj_.append(formats.FramedTransformation(i, j, num, trans))
        return self.corres_traj_

    # ---- CorresApp.cpp:31-110 --------------------------------------------------------------------
    def LoadData(self, filename, num):
        cut = max(filename.rfind("\\"), -1)
        if cut < 0:
            cut = filename.rfind("/")
        self.m_pDirName = filename[:cut + 1]
        if num > 0:
            self.InitialPairs(filename, num)
        else:
            self.corres_traj_ = formats.load_log(filename)
            self.num_ = self.corres_traj_[0].frame
        self.pointclouds_ = [None] * self.num_
        grid_cell = max(self.reg_dist_, self.dist_thresh_)
        for i in range(self.num_):
            fn = "%scloud_bin_%d.pcd" % (self.m_pDirName, i)
            raw = formats.load_pcd(fn)
            xyz = np.stack([raw["x"], raw["y"], raw["z"]], axis=1).astype(np.float32)
            nrm = np.stack([raw["normal_x"], raw["normal_y"], raw["normal_z"]], axis=1).astype(np.float32)
            keep = ~np.isnan(nrm[:, 0])                                   # :94-98
            xyz, nrm = xyz[keep], nrm[keep]
            self.pointclouds_[i] = Cloud(xyz, nrm, grid_cell, self.device)
            self.pointclouds_[i].host_xyz, self.pointclouds_[i].host_nrm = xyz, nrm
            if self.save_xyzn_:                                           # :100-108
                with open("%scloud_bin_xyzn_%d.xyzn" % (self.m_pDirName, i), "w") as f:
                    for p, q in zip(xyz, nrm):
                        f.write("%.6f %.6f %.6f %.6f %.6f %.6f\n" % (p[0], p[1], p[2], q[0], q[1], q[2]))

    def SetClouds(self, clouds, pairs):
        """In-memory alternative to LoadData for tests/bench: clouds = [Cloud], pairs = [FramedTransformation]."""
        self.pointclouds_ = list(clouds)
        self.num_ = len(clouds)
        self.corres_traj_ = list(pairs)
        self.save_corres_ = False

    # ---- CorresApp.cpp:330-359 -------------------------------------------------------------------
    def Blacklist(self, filename):
        self.blacklist_ = set()
        if os.path.exists(filename):
            with open(filename) as f:
                for line in f:
                    if len(line) > 0 and line[0] != "#" and line.strip():
                        self.blacklist_.add(int(line.split()[0]))

    def GetReduxIndex(self, i, j):
        return i + j * self.num_

    def Redux(self, filename):
        self.redux_ = True
        self.redux_traj_ = formats.load_log(filename)
        self.redux_map_ = {}
        for i, t in enumerate(self.redux_traj_):
            self.redux_map_.setdefault(self.GetReduxIndex(t.id1, t.id2), i)

    # ---- CorresApp.cpp:212-319 -------------------------------------------------------------------
    def Registration(self):
        """Both steps run over the whole pair list at once (the reference's loop is an OpenMP parallel for):
        pre-check of every live pair, then ICP of every accepted pair."""
        self.registration_ = True
        live = []
        for ft in self.corres_traj_:
            if ft.id1 in self.blacklist_ or ft.id2 in self.blacklist_:
                ft.frame = -1
                continue
            if ft.frame == -1:
                continue
            live.append(ft)
        P = self.pointclouds_
        cnts = count_inliers_batch([P[ft.id2] for ft in live], [P[ft.id1] for ft in live], [ft.T for ft in live], self.reg_dist_)   # :249-264
        todo = []
        for ft, cnt in zip(live, cnts):
            cnt = int(cnt)
            pcd0, pcd1 = P[ft.id1], P[ft.id2]
            r1 = float(cnt) / float(max(len(pcd0), 1)) if len(pcd0) else float("inf")
            r2 = float(cnt) / float(max(len(pcd1), 1)) if len(pcd1) else float("inf")
            accept = cnt >= self.reg_num_ or (r1 > self.reg_ratio_ and r2 > self.reg_ratio_)   # :267
            if self.verbose:
                print("    <%d, %d> : %d inliers with ratio %.2f(%d) and %.2f(%d) ... %s" % (
                    ft.id1, ft.id2, cnt, r1, len(pcd0), r2, len(pcd1), "accept." if accept else "reject."))
            if not accept:
                ft.frame = -1
                continue
            ft.frame = cnt
            if self.redux_:
                it = self.redux_map_.get(self.GetReduxIndex(ft.id1, ft.id2))
                if it is not None:
                    ft.T = self.redux_traj_[it].T.copy()
                    continue
            todo.append(ft)
        finals, iters, _, _ = icp_align_batch([P[ft.id2] for ft in todo], [P[ft.id1] for ft in todo],
                                              [ft.T.astype(np.float32) for ft in todo], self.reg_dist_, 20, 1e-6, self.stop_rule)   # :295-306
        for ft, final, it in zip(todo, finals, iters):
            ft.T = final.astype(np.float64)                                              # :312
            self.icp_iterations_[(ft.id1, ft.id2)] = int(it)

    # ---- CorresApp.cpp:112-210 -------------------------------------------------------------------
    def FindCorrespondence(self):
        if self.output_information_:
            self.corres_info_ = [formats.FramedInformation(t.id1, t.id2, t.frame, np.zeros((6, 6))) for t in self.corres_traj_]
        self.correspondences_ = {}
        live = [(idx, ft) for idx, ft in enumerate(self.corres_traj_)
                if not (ft.id1 in self.blacklist_ or ft.id2 in self.blacklist_) and ft.frame != -1]
        P = self.pointclouds_
        for c0 in range(0, len(live), 64):                       # chunks bound the page-locked result arena
            chunk = live[c0:c0 + 64]
            lists, infos = find_correspondence_batch([P[ft.id2] for _, ft in chunk], [P[ft.id1] for _, ft in chunk],
                                                     [ft.T for _, ft in chunk], self.dist_thresh_, self.normal_thresh_,
                                                     self.output_information_, copy=False)
            for k, (idx, ft) in enumerate(chunk):
                corres = lists[k]
                n = corres.shape[0]
                ratio = float(n) / float(ft.frame) if ft.frame != 0 else float("inf")
                if ratio < 0.5:                                                          # :164-171
                    ft.frame = -1 if self.reg_num_ > 0 else n
                else:
                    ft.frame = n
                if self.save_corres_:                                                    # :175-184
                    formats.save_corres("%scorres_%d_%d.txt" % (self.m_pDirName, ft.id1, ft.id2), corres)
                if self.keep_correspondences_:
                    self.correspondences_[(ft.id1, ft.id2)] = corres.copy()
                if self.output_information_:                                             # :186-208
                    self.corres_info_[idx].frame = ft.frame
                    self.corres_info_[idx].info = infos[k]

    # ---- CorresApp.cpp:321-328 -------------------------------------------------------------------
    def Finalize(self):
        formats.save_log(os.path.join(self.out_dir, "reg_output.log"), self.corres_traj_)
        if self.output_information_:
            formats.save_info(os.path.join(self.out_dir, "reg_output.info"), self.corres_info_)
