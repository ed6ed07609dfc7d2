"""Host-side mirror of the data-parallel half of the reference's FragmentOptimizer over the C ABI.

  FragmentOptimizer  <->  COptApp's point clouds + correspondences (FragmentOptimizer/OptApp.h:39-124, PointCloud.h):
      InitPointClouds / InitCorrespondences / UpdatePose / UpdateAllPointPN and the Hessian assembly of OptimizeRigid
      (OptApp.cpp:312-375) and OptimizeSLAC (OptApp.cpp:473-560) run in the HIP kernels of liber_hip.so (er_fopt.hip).
      The sparse solve (CHOLMOD in the reference) and the lattice regularizer stay on the host; OptimizeRigid below
      closes the loop with a dense numpy solve so the tests can check convergence end to end.
"""
import ctypes as C

import numpy as np

from . import _ffi


class Lattice:
    """The control lattice of COptApp and the host-side pieces defined on it (a few thousand vertices: plain numpy, no GPU):
    vertex indexing, the regularizer's edge list and Laplacian, the local rotation fit, pose increments."""

    def __init__(self, resolution=8, length=3.0):
        self.resolution_, self.length_ = int(resolution), float(length)
        self.nper_ = (self.resolution_ + 1) ** 3 * 3

    def GetIndex(self, i, j, k):
        n1 = self.resolution_ + 1
        return i + j * n1 + k * n1 * n1

    def edges(self):
        """(vertex, neighbour) pairs in the order of the reference's six `if` blocks (OptApp.cpp:227-245, 576-611, 773-798)."""
        r, out = self.resolution_, []
        for i in range(r + 1):
            for j in range(r + 1):
                for k in range(r + 1):
                    nb = []
                    if i > 0: nb.append(self.GetIndex(i - 1, j, k))
                    if i < r: nb.append(self.GetIndex(i + 1, j, k))
                    if j > 0: nb.append(self.GetIndex(i, j - 1, k))
                    if j < r: nb.append(self.GetIndex(i, j + 1, k))
                    if k > 0: nb.append(self.GetIndex(i, j, k - 1))
                    if k < r: nb.append(self.GetIndex(i, j, k + 1))
                    out.append((self.GetIndex(i, j, k), nb, (i, j, k)))
        return out

    def laplacian(self):
        """Sum over (vertex, neighbour) of AddHessian2( {v, nb}, {1, -1} ): [[1, -1], [-1, 1]] per xyz component
        (HashSparseMatrix.cpp:50-58) -- every undirected edge is visited from both ends.  Dense nper x nper."""
        L = np.zeros((self.nper_, self.nper_))
        for v, nb, _ in self.edges():
            for w in nb:
                for c in range(3):
                    a, b = v * 3 + c, w * 3 + c
                    L[a, a] += 1.0
                    L[b, b] += 1.0
                    L[a, b] -= 1.0
                    L[b, a] -= 1.0
        return L

    @staticmethod
    def GetRotation(dif, diff):
        """COptApp::GetRotation, OptApp.cpp:850-871: C = sum dif^T diff over the neighbours, R = V U^T (det fixed)."""
        Cm = dif.T @ diff
        U, _, Vt = np.linalg.svd(Cm)
        V = Vt.T
        R = V @ U.T
        if np.linalg.det(R) < 0:
            U = U.copy()
            U[:, 2] *= -1
            R = V @ U.T
        return R

    @staticmethod
    def increment(x6):
        """AngleAxis(z) * AngleAxis(y) * AngleAxis(x) and the translation of one 6-vector of the solution (OptApp.cpp:395-400, 645-651)."""
        a, b, g = x6[:3]
        Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        Rz = np.array([[np.cos(g), -np.sin(g), 0], [np.sin(g), np.cos(g), 0], [0, 0, 1]])
        aff = np.eye(4)
        aff[:3, :3] = Rz @ Ry @ Rx
        aff[:3, 3] = x6[3:6]
        return aff

    def canonical(self):
        """(i, j, k) * unit_length_ per vertex, xyz interleaved (InitCtrSLAC, OptApp.cpp:723-733)."""
        ul = self.length_ / self.resolution_
        out = np.zeros(self.nper_)
        for v, _, (i, j, k) in self.edges():
            out[v * 3:v * 3 + 3] = (i * ul, j * ul, k * ul)
        return out

    @staticmethod
    def apply(P, xyz):
        """Matrix4d * (x, y, z, 1), row by row ((m0 x + m1 y) + m2 z) + m3."""
        return ((P[:3, 0] * xyz[:, :1] + P[:3, 1] * xyz[:, 1:2]) + P[:3, 2] * xyz[:, 2:3]) + P[:3, 3]


class FragmentOptimizer:
    def __init__(self, num, resolution=8, length=3.0, device=0):
        self._lib = _ffi.lib()
        self.num_, self.resolution_, self.length_ = int(num), int(resolution), float(length)
        self.nper_ = (resolution + 1) ** 3 * 3
        self.lattice = Lattice(resolution, length)
        h = C.c_void_p()
        _ffi.check(self._lib.er_fopt_create(self.num_, self.resolution_, C.c_float(length), int(device), C.byref(h)), "er_fopt_create")
        self._h = h
        self.n_pairs = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.er_fopt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- COptApp::InitPointClouds, OptApp.cpp:74-98 ------------------------------------------------
    def SetCloud(self, frag, xyz, normals):
        """Returns -1, or the index of the first point outside the cube (loading stops there, PointCloud.cpp:57-60)."""
        x = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        n = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
        bad = C.c_int(-1)
        _ffi.check(self._lib.er_fopt_set_cloud(self._h, int(frag), _ffi.ptr(x), _ffi.ptr(n), x.shape[0], C.byref(bad)), "er_fopt_set_cloud")
        return bad.value

    def points(self, frag):
        m = self._lib.er_fopt_cloud_size(self._h, int(frag))
        idx0, val, nval = np.zeros(m, np.int32), np.zeros((m, 8), np.float32), np.zeros((m, 8), np.float32)
        p, n = np.zeros((m, 3), np.float32), np.zeros((m, 3), np.float32)
        _ffi.check(self._lib.er_fopt_get_points(self._h, int(frag), _ffi.ptr(idx0), _ffi.ptr(val), _ffi.ptr(nval), _ffi.ptr(p), _ffi.ptr(n)),
                   "er_fopt_get_points")
        return dict(idx0=idx0, val=val, nval=nval, p=p, n=n)

    # ---- PointCloud::UpdatePose / UpdateAllPointPN ------------------------------------------------------
    def UpdatePose(self, frag, M):
        Mm = np.ascontiguousarray(M, np.float32).reshape(16)
        _ffi.check(self._lib.er_fopt_update_pose(self._h, int(frag), _ffi.ptr(Mm)), "er_fopt_update_pose")

    def UpdateAllPointPN(self, expand_ctr):
        c = np.ascontiguousarray(expand_ctr, np.float64).reshape(self.num_, self.nper_)
        for f in range(self.num_):
            row = np.ascontiguousarray(c[f])
            _ffi.check(self._lib.er_fopt_update_point_pn(self._h, f, _ffi.ptr(row)), "er_fopt_update_point_pn")

    def UpdateAllNormal(self, ctr):
        """pointclouds_[l].UpdateAllNormal(ctr) for every fragment (non-rigid mode, OptApp.cpp:151-153); ctr has num * nper entries."""
        c = np.ascontiguousarray(ctr, np.float64).reshape(self.num_, self.nper_)
        for f in range(self.num_):
            row = np.ascontiguousarray(c[f])
            _ffi.check(self._lib.er_fopt_update_normals(self._h, f, _ffi.ptr(row)), "er_fopt_update_normals")

    # ---- COptApp::InitCorrespondences, OptApp.cpp:100-118 ---------------------------------------------
    def SetCorrespondences(self, pairs):
        """pairs: list of (i, j, int32 [m,2] rows (index in fragment i, index in fragment j))."""
        n = len(pairs)
        fi = np.array([p[0] for p in pairs], np.int32)
        fj = np.array([p[1] for p in pairs], np.int32)
        arrs = [np.ascontiguousarray(p[2], np.int32).reshape(-1, 2) for p in pairs]
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrs])
        cnt = np.array([a.shape[0] for a in arrs], np.int32)
        _ffi.check(self._lib.er_fopt_set_correspondences(self._h, n, _ffi.ptr(fi), _ffi.ptr(fj), ptrs, _ffi.ptr(cnt)), "er_fopt_set_correspondences")
        self.n_pairs = n
        return self._lib.er_fopt_group_count(self._h)

    def SetCorrespondencesDev(self, pair_ids, lists):
        """The same from lists that are ALREADY IN HBM (icp.DeviceLists, filled by registration_batch_dev): pair_ids = [(i, j)] with i the
        TARGET fragment of list k and j its source (the rows are (target index, source index), the lines of corres_<i>_<j>.txt).  The sort by
        lattice cell pair runs on the GPU; results are bit-identical to SetCorrespondences on the downloaded lists."""
        n = len(pair_ids)
        fi = np.array([p[0] for p in pair_ids], np.int32)
        fj = np.array([p[1] for p in pair_ids], np.int32)
        cnt = np.ascontiguousarray(lists.counts[:n], np.int32)
        _ffi.check(self._lib.er_fopt_set_correspondences_dev(self._h, n, _ffi.ptr(fi), _ffi.ptr(fj), lists.ptrs(), _ffi.ptr(cnt)),
                   "er_fopt_set_correspondences_dev")
        self.n_pairs = n
        return self._lib.er_fopt_group_count(self._h)

    # ---- Hessian assembly ---------------------------------------------------------------------------------
    def AssembleRigid(self):
        N = 6 * self.num_
        JJ, Jb, sc = np.zeros((N, N)), np.zeros(N), C.c_double(0)
        _ffi.check(self._lib.er_fopt_assemble_rigid(self._h, _ffi.ptr(JJ), _ffi.ptr(Jb), C.byref(sc)), "er_fopt_assemble_rigid")
        return JJ, Jb, sc.value

    def AssembleSLAC(self, pose_rot_t):
        N = 6 * self.num_ + self.nper_
        R = np.ascontiguousarray(pose_rot_t, np.float64).reshape(self.num_, 9)
        JJ, Jb, sc = np.zeros((N, N)), np.zeros(N), C.c_double(0)
        _ffi.check(self._lib.er_fopt_assemble_slac(self._h, _ffi.ptr(R), _ffi.ptr(JJ), _ffi.ptr(Jb), C.byref(sc)), "er_fopt_assemble_slac")
        return JJ, Jb, sc.value

    def AssembleNonrigid(self, weight):
        """Data term of OptimizeNonrigid (OptApp.cpp:159-206) as block-sparse arrays:
        diag [num, (res+1)^3, 24, 24], offdiag [groups, 24, 24], info [groups, 4] = (frag_i, frag_j, idx0_i, idx0_j)."""
        nv = (self.resolution_ + 1) ** 3
        ng = self._lib.er_fopt_group_count(self._h)
        diag = np.zeros((self.num_, nv, 24, 24))
        off = np.zeros((max(ng, 1), 24, 24))
        info = np.zeros((max(ng, 1), 4), np.int32)
        _ffi.check(self._lib.er_fopt_group_info(self._h, _ffi.ptr(info)), "er_fopt_group_info")
        _ffi.check(self._lib.er_fopt_assemble_nonrigid(self._h, float(weight), _ffi.ptr(diag), _ffi.ptr(off)), "er_fopt_assemble_nonrigid")
        return diag, off[:ng], info[:ng]

    def local_to_lattice(self):
        """Lattice offset of local bucket entry c*8 + t relative to idx_[0]: vertex_offset(t) + c (PointCloud.h:113-120)."""
        n1 = self.resolution_ + 1
        t = np.arange(8)
        voff = (((t >> 2) & 1) + ((t >> 1) & 1) * n1 + (t & 1) * n1 * n1) * 3
        return (voff[None, :] + np.arange(3)[:, None]).reshape(24)

    def NonrigidTriplets(self, weight):
        """thisAA - baseAA as merged COO triplets (rows, cols, vals) with global indices fragment * nper + lattice index --
        what the reference hands to CHOLMOD after adding baseAA."""
        diag, off, info = self.AssembleNonrigid(weight)
        loc = self.local_to_lattice()
        M = self.nper_ * self.num_
        f, v = np.nonzero(np.abs(diag).reshape(self.num_, diag.shape[1], -1).max(-1) > 0)
        base = f * self.nper_ + v * 3
        r = (base[:, None] + loc[None, :])
        keys = [(r[:, :, None] * M + r[:, None, :]).reshape(-1)]
        vals = [diag[f, v].reshape(-1)]
        if off.shape[0]:
            ri = info[:, 0].astype(np.int64) * self.nper_ + info[:, 2]
            rj = info[:, 1].astype(np.int64) * self.nper_ + info[:, 3]
            a = ri[:, None] + loc[None, :]
            b = rj[:, None] + loc[None, :]
            keys.append((a[:, :, None] * M + b[:, None, :]).reshape(-1))
            vals.append(off.reshape(-1))
        keys, vals = np.concatenate(keys), np.concatenate(vals)
        uk, inv = np.unique(keys, return_inverse=True)
        out = np.zeros(uk.size)
        np.add.at(out, inv, vals)
        return uk // M, uk % M, out

    # ---- systems kept and solved on the device (own blocked Cholesky in HBM over rocBLAS level-3 calls) -----------
    def FactorSLAC(self, pose_rot_t, default_weight):
        """thisJJ of one OptimizeSLAC iteration assembled and factored on the device.  Returns (dataJb, data score)."""
        N = 6 * self.num_ + self.nper_
        R = np.ascontiguousarray(pose_rot_t, np.float64).reshape(self.num_, 9)
        Jb, sc = np.zeros(N), C.c_double(0)
        _ffi.check(self._lib.er_fopt_factor_slac(self._h, _ffi.ptr(R), float(default_weight), _ffi.ptr(Jb), C.byref(sc)), "er_fopt_factor_slac")
        return Jb, sc.value

    def FactorNonrigid(self, weight):
        _ffi.check(self._lib.er_fopt_factor_nonrigid(self._h, float(weight)), "er_fopt_factor_nonrigid")

    def DebugShiftDiagonal(self, index, value):
        """Test hook (er_fopt_debug_shift_diagonal): add `value` to diagonal entry `index` of every system factored from now on; index < 0 = off."""
        _ffi.check(self._lib.er_fopt_debug_shift_diagonal(self._h, int(index), float(value)), "er_fopt_debug_shift_diagonal")

    def Solve(self, rhs, add_data_jb=False):
        b = np.ascontiguousarray(rhs, np.float64).reshape(-1)
        x = np.zeros_like(b)
        _ffi.check(self._lib.er_fopt_solve(self._h, _ffi.ptr(b), 1 if add_data_jb else 0, _ffi.ptr(x)), "er_fopt_solve")
        return x

    # ---- host-side pieces of COptApp: see class Lattice ----------------------------------------------------------
    def GetIndex(self, i, j, k):
        return self.lattice.GetIndex(i, j, k)

    def _lattice_edges(self):
        return self.lattice.edges()

    def _laplacian(self):
        return self.lattice.laplacian()

    def _canonical_lattice(self):
        return self.lattice.canonical()

    GetRotation = staticmethod(lambda dif, diff: Lattice.GetRotation(dif, diff))
    _increment = staticmethod(lambda x6: Lattice.increment(x6))
    _apply = staticmethod(lambda P, xyz: Lattice.apply(P, xyz))

    # ---- COptApp::OptimizeSLAC, OptApp.cpp:414-680 ----------------------------------------------------------------
    def OptimizeSLAC(self, ipose, weight=1.0, max_iteration=5, solver="device"):
        """Returns (poses, expand_ctr [num * nper], data scores).  The data term comes from er_fopt_assemble_slac; base term,
        regularizer, dense solve (CHOLMOD in the reference) and the pose / lattice updates follow the reference line by line."""
        num, nper = self.num_, self.nper_
        N = 6 * num + nper
        default_weight = num * weight                                                # :416
        pose = [np.array(P, np.float64) for P in ipose]
        ictr = self._canonical_lattice()
        thisCtr = ictr.copy()
        for l in range(num):
            self.UpdatePose(l, pose[l].astype(np.float32))                           # :443
        base = np.zeros((N, N))
        base[6 * num:, 6 * num:] = self._laplacian()
        anchor = 6 * num + self.GetIndex(self.resolution_ // 2, self.resolution_ // 2, 0) * 3
        for c in range(3):
            base[anchor + c, anchor + c] += 1.0                                      # :839-843
        base *= default_weight                                                       # :452
        edges = self._lattice_edges()
        scores = []
        for _ in range(max_iteration):
            Rt = np.stack([P[:3, :3].T.reshape(9) for P in pose])
            if solver == "device":                                                   # system assembled, factored and solved in HBM
                dataJb, score = self.FactorSLAC(Rt, default_weight)
            else:
                JJ, dataJb, score = self.AssembleSLAC(Rt)
                thisJJ = np.triu(base) + JJ                                          # :456 (Upper view) + data term
                thisJJ[np.arange(6), np.arange(6)] += 1.0                            # :459-464
                full = thisJJ + np.triu(thisJJ, 1).T
            scores.append(score)
            baseJb = np.zeros(N)
            cur, ini = thisCtr.reshape(-1, 3), ictr.reshape(-1, 3)
            for v, nb, ijk in edges:                                                 # regularizer, :570-631
                dif = ini[v] - ini[nb]
                diff = cur[v] - cur[nb]
                R = np.eye(3) if ijk == (self.resolution_ // 2, self.resolution_ // 2, 0) else self.GetRotation(dif, diff)
                bx = (diff - dif @ R.T) * default_weight
                baseJb[6 * num + v * 3:6 * num + v * 3 + 3] += bx.sum(0)
                for t, w in enumerate(nb):
                    baseJb[6 * num + w * 3:6 * num + w * 3 + 3] -= bx[t]
            result = -(self.Solve(dataJb + baseJb) if solver == "device" else np.linalg.solve(full, dataJb + baseJb))   # :632-638
            thisCtr = thisCtr + result[6 * num:]                                     # :644-646
            for l in range(num):
                pose[l] = self._increment(result[l * 6:l * 6 + 6]) @ pose[l]         # :648-658
            expand = np.concatenate([self._apply(pose[l], thisCtr.reshape(-1, 3)).reshape(-1) for l in range(num)])   # ExpandCtr, :752-763
            self.UpdateAllPointPN(expand)                                            # :660-663
        expand = np.concatenate([self._apply(pose[l], thisCtr.reshape(-1, 3)).reshape(-1) for l in range(num)])
        return pose, expand, scores

    # ---- COptApp::OptimizeNonrigid, OptApp.cpp:120-278 ------------------------------------------------------------
    def OptimizeNonrigid(self, ipose, weight=1.0, max_iteration=5, max_inner_iteration=10, solver="device"):
        """Returns (ctr [num * nper], inner-iteration scores).  Data term from er_fopt_assemble_nonrigid; regularizer, dense
        solve and control flow as in the reference."""
        num, nper = self.num_, self.nper_
        M = num * nper
        lat = self._canonical_lattice().reshape(-1, 3)
        ctr = np.concatenate([self._apply(np.array(P, np.float64), lat).reshape(-1) for P in ipose])    # InitCtr, :709-721
        ictr = ctr.copy()
        if solver != "device":
            baseAA = np.zeros((M, M))
            Lp = self._laplacian()
            for l in range(num):
                baseAA[l * nper:(l + 1) * nper, l * nper:(l + 1) * nper] = Lp
            for c in range(3):
                baseAA[c, c] += 1.0                                                  # :803-807
        edges = self._lattice_edges()
        scores = []
        for _ in range(max_iteration):
            self.UpdateAllNormal(ctr)                                                # :151-153
            if solver == "device":
                self.FactorNonrigid(weight)
            else:
                r, c, v = self.NonrigidTriplets(weight)
                thisAA = baseAA.copy()
                np.add.at(thisAA, (r, c), v)
                thisAA = np.triu(thisAA) + np.triu(thisAA, 1).T                      # the solver reads the Upper triangle
            for _m in range(max_inner_iteration):
                Ab = np.zeros(M)
                for l in range(num):
                    cur = ctr[l * nper:(l + 1) * nper].reshape(-1, 3)
                    ini = ictr[l * nper:(l + 1) * nper].reshape(-1, 3)
                    for vv, nb, _ in edges:                                          # :221-260
                        dif = ini[vv] - ini[nb]
                        R = self.GetRotation(dif, cur[vv] - cur[nb])
                        bx = dif @ R.T
                        Ab[l * nper + vv * 3:l * nper + vv * 3 + 3] += bx.sum(0)
                        for t, w in enumerate(nb):
                            Ab[l * nper + w * 3:l * nper + w * 3 + 3] -= bx[t]
                old = ctr
                ctr = self.Solve(Ab) if solver == "device" else np.linalg.solve(thisAA, Ab)   # :263
                scores.append(float(np.linalg.norm(old - ctr)))
        return ctr, scores

    # ---- COptApp::OptimizeRigid, OptApp.cpp:282-412 (dense numpy solve in place of CHOLMOD) ----------------
    def OptimizeRigid(self, ipose, max_iteration=5):
        """ipose: list of float64 4x4 initial poses.  Returns (poses, scores per iteration)."""
        pose = [np.array(P, np.float64) for P in ipose]
        for l in range(self.num_):
            self.UpdatePose(l, pose[l].astype(np.float32))                       # :296
        scores = []
        for _ in range(max_iteration):
            JJ, Jb, score = self.AssembleRigid()
            scores.append(score)
            result = -np.linalg.solve(JJ, Jb)                                        # solver.solve( thisJb ), :389-393
            for l in range(self.num_):
                a, b, g = result[l * 6:l * 6 + 3]
                Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
                Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
                Rz = np.array([[np.cos(g), -np.sin(g), 0], [np.sin(g), np.cos(g), 0], [0, 0, 1]])
                aff = np.eye(4)
                aff[:3, :3] = Rz @ Ry @ Rx                                           # AngleAxis Z * Y * X, :396-399
                aff[:3, 3] = result[l * 6 + 3:l * 6 + 6]
                pose[l] = aff @ pose[l]                                              # :401
                self.UpdatePose(l, aff.astype(np.float32))                           # :402
        return pose, scores
