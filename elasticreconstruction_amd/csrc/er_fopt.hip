// er_fopt.hip -- SURVEY.md 8f-2: the data-parallel half of the reference's FragmentOptimizer on MI355X (gfx950):
// per-point state (PointCloud.h:6-176), UpdatePose / UpdateAllPointPN, and the Hessian assembly of OptimizeRigid
// (OptApp.cpp:312-375) and OptimizeSLAC (OptApp.cpp:473-560).  The CHOLMOD solve and the lattice regularizer (a few
// thousand vertices) stay with the host, as in the reference.
//
// Assembly = grouped Gram matrices on the FP64 matrix cores.  Every correspondence contributes v v^T (+ b v, b^2) where
// v is a "bucket" of 12 (rigid) or 60 (SLAC: 12 pose + 24 + 24 lattice) values whose matrix indices depend only on the
// fragment pair and on the two control-lattice cells the points fall in.  Correspondences are therefore sorted ONCE
// (they and the cells never change during the optimisation) by (pair, cell of p_i, cell of p_j); one wave per group
// chunk accumulates G = sum_k [v_k; b_k][v_k; b_k]^T with v_mfma_f64_16x16x4_f64 -- lane (r, q) computes entry
// 16*blk + r of the bucket of correspondence k0 + q, and that register is at once the A operand of block row blk and
// the B operand of block column blk, so the 61-vector never touches LDS -- and then adds the upper triangle of G into
// the dense matrix with float64 atomics, folding coinciding lattice indices exactly like OptApp.cpp:537-548.
// This is the one GEMM-shaped loop of the pipeline; the reference runs it as scalar double loops under OpenMP.
#include "er_common.h"

#include "../../include/er_hip.h"

#include <dlfcn.h>
#include <hipcub/hipcub.hpp>   // radix sort / run-length encode of er_fopt_set_correspondences_dev (library primitives; the assembly kernels are hand-written)

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int kBlock = 256;
#ifndef ER_FOPT_CHUNK
#define ER_FOPT_CHUNK 512
#endif
#ifndef ER_FOPT_MINBLOCKS
#define ER_FOPT_MINBLOCKS 2
#endif
constexpr int kChunkMax = ER_FOPT_CHUNK;   // correspondences per wave task

typedef double double4_t __attribute__((ext_vector_type(4)));

struct FragPtr {
  const int* idx0;
  const float* val;
  float* p;
  float* nrm;
};

struct Chunk {
  int fi, fj;        // fragments (corres_.idx0_, idx1_)
  int ci, cj;        // idx_[0] of the lattice cell of p_i / p_j (vertex index * 3)
  int start, count;  // range in the sorted correspondence arrays
  int group;         // index of the (pair, cell, cell) group this chunk belongs to
};

// vertex offsets of idx_[0..7] relative to idx_[0], PointCloud.h:113-120 (t = 4*dx + 2*dy + dz)
__host__ __device__ inline int vertex_offset(int t, int res) {
  const int n1 = res + 1;
  return (((t >> 2) & 1) + ((t >> 1) & 1) * n1 + (t & 1) * n1 * n1) * 3;
}

// PointCloud::UpdatePose, PointCloud.h:71-83
__global__ void k_fopt_update_pose(float* __restrict__ p, float* __restrict__ nrm, int n, const float* __restrict__ M) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const float x = p[3 * k], y = p[3 * k + 1], z = p[3 * k + 2];
  const float a = nrm[3 * k], b = nrm[3 * k + 1], c = nrm[3 * k + 2];
#pragma unroll
  for (int r = 0; r < 3; r++) {
    p[3 * k + r] = ((M[4 * r] * x + M[4 * r + 1] * y) + M[4 * r + 2] * z) + M[4 * r + 3] * 1.0f;
    nrm[3 * k + r] = ((M[4 * r] * a + M[4 * r + 1] * b) + M[4 * r + 2] * c) + M[4 * r + 3] * 0.0f;
  }
}

// PointCloud::UpdateAllPointPN, PointCloud.h:44-52 (UpdateNormal :58-69, UpdatePoint :85-94); ctr = the fragment's slice
// normals_only = 1: PointCloud::UpdateAllNormal (PointCloud.h:32-36), the non-rigid mode's per-iteration update (OptApp.cpp:151-153)
__global__ void k_fopt_update_pn(const int* __restrict__ idx0, const float* __restrict__ val, const float* __restrict__ nval,
                                 float* __restrict__ p, float* __restrict__ nrm, int n, const double* __restrict__ ctr, int res,
                                 int normals_only) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int base = idx0[k];
  float c[8][3];
#pragma unroll
  for (int t = 0; t < 8; t++) {
    const int o = base + vertex_offset(t, res);
#pragma unroll
    for (int i = 0; i < 3; i++) c[t][i] = (float)ctr[o + i];
  }
  float nn[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; t++) s += nval[8 * k + t] * c[t][i];
    nn[i] = s;
  }
  const float len = sqrtf(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
#pragma unroll
  for (int i = 0; i < 3; i++) nrm[3 * k + i] = nn[i] / len;
  if (normals_only) return;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double pos = 0.0;
#pragma unroll
    for (int t = 0; t < 8; t++) pos += (double)(val[8 * k + t] * c[t][i]);
    p[3 * k + i] = (float)pos;
  }
}

// MODE 0 = rigid (12 entries + b at 12, one 16x16 tile), MODE 1 = SLAC (60 entries + b at 60, 4x4 blocks, upper 10 tiles),
// MODE 2 = non-rigid (OptApp.cpp:159-206: val1 (24) | val2 (24), 3x3 blocks, upper 6 tiles; no right-hand side: Ab comes from
// the regularizer only).  MODE 2 writes block-sparse output: G11 / G22 into the per-fragment, per-cell 24x24 blocks `JJ`
// ([fragment][corner vertex][24][24], both triangles like AddHessian) and G12 into `Jb` = one 24x24 block per group.
template <int MODE>
__global__ __launch_bounds__(kBlock, ER_FOPT_MINBLOCKS) void k_fopt_gram(const Chunk* __restrict__ chunks, int n_chunks, const FragPtr* __restrict__ frags,
                                                      const int* __restrict__ first, const int* __restrict__ second,
                                                      const double* __restrict__ rot_t, int num, int res, int N,
                                                      double* __restrict__ JJ, double* __restrict__ Jb, double* __restrict__ score) {
  constexpr int NB = MODE == 0 ? 1 : (MODE == 1 ? 4 : 3);      // 16-entry blocks of the bucket
  constexpr int NT = NB * (NB + 1) / 2;                        // upper-triangular tile pairs
  constexpr int BPOS = MODE == 0 ? 12 : (MODE == 1 ? 60 : 48); // where b sits (MODE 2: nothing there)
  const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  if (wave >= n_chunks) return;
  const Chunk ch = chunks[wave];
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const FragPtr Fi = frags[ch.fi], Fj = frags[ch.fj];
  double Ri[9], Rj[9];
  if (MODE == 1) {
#pragma unroll
    for (int t = 0; t < 9; t++) {
      Ri[t] = rot_t[9 * ch.fi + t];
      Rj[t] = rot_t[9 * ch.fj + t];
    }
  }
  double4_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};

  for (int k0 = 0; k0 < ch.count; k0 += 4) {
    const int k = k0 + q;
    double a[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) a[b] = 0.0;
    if (k < ch.count) {
      const int ii = first[ch.start + k], jj = second[ch.start + k];
      const double ppi[3] = {Fi.p[3 * ii], Fi.p[3 * ii + 1], Fi.p[3 * ii + 2]};
      const double ppj[3] = {Fj.p[3 * jj], Fj.p[3 * jj + 1], Fj.p[3 * jj + 2]};
      const double npi[3] = {Fi.nrm[3 * ii], Fi.nrm[3 * ii + 1], Fi.nrm[3 * ii + 2]};
      const double d[3] = {ppi[0] - ppj[0], ppi[1] - ppj[1], ppi[2] - ppj[2]};
      const double bval = (d[0] * npi[0] + d[1] * npi[1]) + d[2] * npi[2];          // OptApp.cpp:346 / :493
      if (MODE == 0) {                                                              // OptApp.cpp:363-374
        double v;
        if (r == 0) v = (-ppi[2] * npi[1] + ppi[1] * npi[2]) + (-npi[2] * d[1] + npi[1] * d[2]);
        else if (r == 1) v = (ppi[2] * npi[0] - ppi[0] * npi[2]) + (npi[2] * d[0] - npi[0] * d[2]);
        else if (r == 2) v = (-ppi[1] * npi[0] + ppi[0] * npi[1]) + (-npi[1] * d[0] + npi[0] * d[1]);
        else if (r < 6) v = npi[r - 3];
        else if (r == 6) v = -(-ppj[2] * npi[1] + ppj[1] * npi[2]);
        else if (r == 7) v = -(ppj[2] * npi[0] - ppj[0] * npi[2]);
        else if (r == 8) v = -(-ppj[1] * npi[0] + ppj[0] * npi[1]);
        else if (r < 12) v = -npi[r - 9];
        else if (r == 12) v = bval;
        else v = 0.0;
        a[0] = v;
      } else if (MODE == 2) {                                                       // OptApp.cpp:176-190, entry c*8 + t
        const double weight = rot_t[0];
#pragma unroll
        for (int b = 0; b < NB; b++) {
          const int e = b * 16 + r, h = e < 24 ? e : e - 24, c = h >> 3, tt = h & 7;
          const double nc = c == 0 ? npi[0] : (c == 1 ? npi[1] : npi[2]);
          const double w = e < 24 ? (double)Fi.val[8 * ii + tt] : -(double)Fj.val[8 * jj + tt];
          a[b] = (w * weight) * nc;
        }
      } else {                                                                      // OptApp.cpp:509-535
        const double t[3] = {ppj[1] * npi[2] - ppj[2] * npi[1], ppj[2] * npi[0] - ppj[0] * npi[2], ppj[0] * npi[1] - ppj[1] * npi[0]};
        double dTi[3], dTj[3];
#pragma unroll
        for (int x = 0; x < 3; x++) {
          dTi[x] = (Ri[3 * x] * npi[0] + Ri[3 * x + 1] * npi[1]) + Ri[3 * x + 2] * npi[2];
          dTj[x] = -((Rj[3 * x] * npi[0] + Rj[3 * x + 1] * npi[1]) + Rj[3 * x + 2] * npi[2]);
        }
#pragma unroll
        for (int b = 0; b < NB; b++) {
          const int e = b * 16 + r;
          double v;
          if (e < 12) {
            const int x = e % 3;
            const double base = (e % 6) < 3 ? (x == 0 ? t[0] : (x == 1 ? t[1] : t[2])) : (x == 0 ? npi[0] : (x == 1 ? npi[1] : npi[2]));
            v = e < 6 ? base : -base;
          } else if (e < 36) {
            const int ll = (e - 12) / 3, x = (e - 12) % 3;
            v = (double)Fi.val[8 * ii + ll] * (x == 0 ? dTi[0] : (x == 1 ? dTi[1] : dTi[2]));
          } else if (e < 60) {
            const int ll = (e - 36) / 3, x = (e - 36) % 3;
            v = (double)Fj.val[8 * jj + ll] * (x == 0 ? dTj[0] : (x == 1 ? dTj[1] : dTj[2]));
          } else {
            v = e == 60 ? bval : 0.0;
          }
          a[b] = v;
        }
      }
    }
    // G += [v;b][v;b]^T : tile (bi, bj) uses a[bi] as A (A[i = lane%16][k = lane/16]) and a[bj] as B (B[k][j = lane%16])
    int t = 0;
#pragma unroll
    for (int bi = 0; bi < NB; bi++)
#pragma unroll
      for (int bj = bi; bj < NB; bj++) {
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bi], a[bj], acc[t], 0, 0, 0);
        t++;
      }
  }

  // scatter: D[i = q + 4 v][j = r] of tile (bi, bj) is G[16 bi + i][16 bj + j]  (layout probed: scripts/ubench/mfma_f64_layout.hip)
  if (MODE == 2) {
    const int nv = (res + 1) * (res + 1) * (res + 1);
    double* Di = JJ + ((size_t)ch.fi * nv + ch.ci / 3) * 576;
    double* Dj = JJ + ((size_t)ch.fj * nv + ch.cj / 3) * 576;
    double* Oij = Jb + (size_t)ch.group * 576;
    int t = 0;
#pragma unroll
    for (int bi = 0; bi < NB; bi++)
#pragma unroll
      for (int bj = bi; bj < NB; bj++) {
#pragma unroll
        for (int v = 0; v < 4; v++) {
          const int gi = bi * 16 + q + 4 * v, gj = bj * 16 + r;
          const double G = acc[t][v];
          if (gi > gj || G == 0.0) continue;
          if (gj < 24) {                                        // mati.AddHessian( idx1, val1, 24 ): both triangles
            atomicAdd(&Di[gi * 24 + gj], G);
            if (gi != gj) atomicAdd(&Di[gj * 24 + gi], G);
          } else if (gi >= 24) {                                // matj.AddHessian( idx2, val2, 24 )
            atomicAdd(&Dj[(gi - 24) * 24 + (gj - 24)], G);
            if (gi != gj) atomicAdd(&Dj[(gj - 24) * 24 + (gi - 24)], G);
          } else {                                              // matij.AddHessian( idx1, val1, 24, idx2, val2, 24 )
            atomicAdd(&Oij[gi * 24 + (gj - 24)], G);
          }
        }
        t++;
      }
    return;
  }
  const int lat = 6 * num;
  auto index_of = [&](int g) -> int {
    if (g < 6) return ch.fi * 6 + g;
    if (g < 12) return ch.fj * 6 + (g - 6);
    if (g < 36) return lat + ch.ci + vertex_offset((g - 12) / 3, res) + (g - 12) % 3;
    return lat + ch.cj + vertex_offset((g - 36) / 3, res) + (g - 36) % 3;
  };
  int t = 0;
#pragma unroll
  for (int bi = 0; bi < NB; bi++)
#pragma unroll
    for (int bj = bi; bj < NB; bj++) {
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int gi = bi * 16 + q + 4 * v, gj = bj * 16 + r;
        const double G = acc[t][v];
        if (gi > gj || gj > BPOS || G == 0.0) continue;          // lower half of a diagonal tile, padding, nothing to add
        if (gj == BPOS) {
          if (gi == BPOS) atomicAdd(score, G);                                       // sum b^2
          else atomicAdd(&Jb[index_of(gi)], G);                                      // sum b v  (:549 / AddJb)
          continue;
        }
        const int ia = index_of(gi), ic = index_of(gj);
        if (MODE == 0) {                                                             // AddHessian: both triangles
          atomicAdd(&JJ[(size_t)ia * N + ic], G);
          if (gi != gj) atomicAdd(&JJ[(size_t)ic * N + ia], G);
        } else if (gi == gj) {
          atomicAdd(&JJ[(size_t)ia * N + ia], G);                                    // :538
        } else if (ia == ic) {
          atomicAdd(&JJ[(size_t)ia * N + ia], 2.0 * G);                              // :540-541
        } else if (ia < ic) {
          atomicAdd(&JJ[(size_t)ia * N + ic], G);                                    // :542-543
        } else {
          atomicAdd(&JJ[(size_t)ic * N + ia], G);                                    // :544-545
        }
      }
      t++;
    }
}

// ---- the linear systems, kept and solved on the device -----------------------------------------------------------
// Row-major UPPER triangle throughout (= column-major lower triangle for the Cholesky and rocBLAS).
// AddHessian2( {v, w}, {1, -1} ) for every (vertex, neighbour) ordered pair of the lattice (OptApp.cpp:765-800, 811-836):
// +scale on both diagonals, -scale on the coupling, per xyz component.
// The non-rigid system either as ONE dense matrix (base, ld) or as the block-sparse lower triangle of fragment blocks
// (bs x bs blocks of one fragment's lattice each; blk_off[bq * nb + br] = offset of block (bq, br), bq >= br, column-major
// inside the block -- which is the same "row r, column q, r <= q -> base[r * ld + q]" addressing as the dense row-major
// upper triangle, applied inside the block).
struct MatView {
  double* base;
  long ld;
  const long* blk_off;        // nullptr: dense
  int nb;
  long bs;
};
__device__ __forceinline__ double* mat_at(const MatView& V, long r, long q) {      // r <= q
  if (!V.blk_off) return V.base + r * V.ld + q;
  const long br = r / V.bs, bq = q / V.bs;
  return V.base + V.blk_off[bq * V.nb + br] + (r - br * V.bs) * V.bs + (q - bq * V.bs);
}

__global__ void k_fopt_add_laplacian_v(MatView V, long off, int res, double scale) {
  const int n1 = res + 1, nv = n1 * n1 * n1;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nv * 6) return;
  const int v = t / 6, dir = t % 6;
  const int i = v % n1, j = (v / n1) % n1, k = v / (n1 * n1);
  int w = -1;
  if (dir == 0 && i > 0) w = v - 1;
  if (dir == 1 && i < res) w = v + 1;
  if (dir == 2 && j > 0) w = v - n1;
  if (dir == 3 && j < res) w = v + n1;
  if (dir == 4 && k > 0) w = v - n1 * n1;
  if (dir == 5 && k < res) w = v + n1 * n1;
  if (w < 0) return;
  for (int c = 0; c < 3; c++) {
    const long a = off + (long)v * 3 + c, b = off + (long)w * 3 + c;
    atomicAdd(mat_at(V, a, a), scale);
    atomicAdd(mat_at(V, b, b), scale);
    atomicAdd(mat_at(V, a < b ? a : b, a < b ? b : a), -scale);
  }
}

__global__ void k_fopt_add_diag_v(MatView V, long first, int count, double value) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < count) atomicAdd(mat_at(V, first + t, first + t), value);
}

__global__ void k_fopt_add_laplacian(double* __restrict__ A, long ld, long off, int res, double scale) {
  const int n1 = res + 1, nv = n1 * n1 * n1;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nv * 6) return;
  const int v = t / 6, dir = t % 6;
  const int i = v % n1, j = (v / n1) % n1, k = v / (n1 * n1);
  int w = -1;
  if (dir == 0 && i > 0) w = v - 1;
  if (dir == 1 && i < res) w = v + 1;
  if (dir == 2 && j > 0) w = v - n1;
  if (dir == 3 && j < res) w = v + n1;
  if (dir == 4 && k > 0) w = v - n1 * n1;
  if (dir == 5 && k < res) w = v + n1 * n1;
  if (w < 0) return;
  for (int c = 0; c < 3; c++) {
    const long a = off + (long)v * 3 + c, b = off + (long)w * 3 + c;
    atomicAdd(&A[a * ld + a], scale);
    atomicAdd(&A[b * ld + b], scale);
    atomicAdd(&A[(a < b ? a : b) * ld + (a < b ? b : a)], -scale);
  }
}

__global__ void k_fopt_add_diag(double* __restrict__ A, long ld, long first, int count, double value) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < count) atomicAdd(&A[(first + t) * ld + first + t], value);
}

// non-rigid: the 24x24 blocks of er_fopt_assemble_nonrigid scattered into the dense upper triangle
__global__ void k_fopt_scatter_blocks(const double* __restrict__ blocks, long n_blocks, const int* __restrict__ info, int diag_mode, int nv,
                                      int res, long nper, MatView V) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_blocks * 576) return;
  const double v = blocks[t];
  if (v == 0.0) return;
  const long blk = t / 576;
  const int a = (int)(t % 576) / 24, c = (int)(t % 24);
  long bi, bj;
  if (diag_mode) {
    const long l = blk / nv, vert = blk % nv;
    bi = bj = l * nper + vert * 3;
  } else {
    bi = (long)info[blk * 4] * nper + info[blk * 4 + 2];
    bj = (long)info[blk * 4 + 1] * nper + info[blk * 4 + 3];
  }
  const long r = bi + vertex_offset(a & 7, res) + (a >> 3), q = bj + vertex_offset(c & 7, res) + (c >> 3);
  if (diag_mode) {
    if (r <= q) atomicAdd(mat_at(V, r, q), v);                // both triangles are present in the block: keep the upper one
  } else {
    atomicAdd(mat_at(V, r < q ? r : q, r < q ? q : r), v);
  }
}

__global__ void k_fopt_axpy(double* __restrict__ y, const double* __restrict__ x, long n) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) y[t] += x[t];
}

// rocBLAS (level-3 BLAS for the trailing updates and triangular solves of the Cholesky factorisation below) is loaded on first use:
// the TSDF and ICP paths never pay for it.  Round 3 dropped rocSOLVER: its potrf reported non-positive pivots for positive definite
// matrices under multi-process load (round 2 retried around it); the factorisation is now potrf_lower below -- an own 64 x 64
// diagonal-block kernel + rocblas_dtrsm / dsyrk -- and potrs is two rocblas_dtrsv.  Caveat met on this ROCm: when the host PROGRAM
// reached HIP through its start-up dependencies (liber_hip.so as DT_NEEDED), a later dlopen of a ROCm math library can hang in its
// static initialisation; such programs link librocblas themselves (bin/FragmentOptimizer does), after which the dlopen below only
// finds the library already loaded.  From Python (ctypes) the lazy load works as is.
struct RocBlas {
  void *blas = nullptr, *handle = nullptr;
  int (*create_handle)(void**) = nullptr;
  int (*destroy_handle)(void*) = nullptr;
  int (*set_stream)(void*, hipStream_t) = nullptr;
  // block-sparse factorisation of the non-rigid system: level-3 BLAS on fragment blocks
  int (*dtrsm)(void*, int, int, int, int, int, int, const double*, const double*, int, double*, int) = nullptr;
  int (*dgemm)(void*, int, int, int, int, int, const double*, const double*, int, const double*, int, const double*, double*, int) = nullptr;
  int (*dsyrk)(void*, int, int, int, int, const double*, const double*, int, const double*, double*, int) = nullptr;
  int (*dgemv)(void*, int, int, int, const double*, const double*, int, const double*, int, const double*, double*, int) = nullptr;
  int (*dtrsv)(void*, int, int, int, int, const double*, int, double*, int) = nullptr;
};
constexpr int kFillLower = 122;                                 // rocblas_fill_lower
constexpr int kOpN = 111, kOpT = 112, kDiagNonUnit = 131, kSideRight = 142;   // rocblas_operation / diagonal / side

// Cholesky factorisation of one diagonal block (jb <= 64, column-major, lower triangle) by ONE workgroup in LDS: right-looking, column by
// column -- pivot, scale the column, rank-1 update of the trailing triangle.  A non-positive (or NaN) pivot records its 1-based global
// index in *info (LAPACK's convention; the first one wins) and the block is left as it is; a launch that finds *info set returns at once.
constexpr int kPotrfNb = 64;
__global__ __launch_bounds__(256) void k_potrf_block(double* __restrict__ A, long lda, int jb, int j0, int* __restrict__ info) {
  __shared__ double a[kPotrfNb][kPotrfNb + 1];
  if (*info != 0) return;
  const int tid = threadIdx.x;
  for (int t = tid; t < jb * jb; t += 256) {
    const int r = t % jb, c = t / jb;
    a[r][c] = r >= c ? A[(size_t)c * lda + r] : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < jb; j++) {
    const double d = a[j][j];
    if (!(d > 0.0)) {                                        // uniform: every thread reads the same LDS word
      if (tid == 0) atomicCAS(info, 0, j0 + j + 1);
      return;
    }
    const double l = sqrt(d);
    __syncthreads();                                           // everybody has read a[j][j]
    if (tid == 0) a[j][j] = l;
    for (int r = j + 1 + tid; r < jb; r += 256) a[r][j] = a[r][j] / l;
    __syncthreads();
    // trailing update of the lower triangle: a[r][c] -= a[r][j] a[c][j] for j < c <= r
    const int m = jb - j - 1;
    for (int t = tid; t < m * m; t += 256) {
      const int r = j + 1 + t % m, c = j + 1 + t / m;
      if (r >= c) a[r][c] -= a[r][j] * a[c][j];
    }
    __syncthreads();
  }
  for (int t = tid; t < jb * jb; t += 256) {
    const int r = t % jb, c = t / jb;
    if (r >= c) A[(size_t)c * lda + r] = a[r][c];
  }
}

int rocblas_load(RocBlas& R, hipStream_t stream) {
  if (R.handle) return 0;
  R.blas = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
  if (!R.blas) R.blas = dlopen("/opt/rocm/lib/librocblas.so", RTLD_NOW | RTLD_GLOBAL);
  if (!R.blas) return er::fail("the on-device solve needs librocblas.so: %s", dlerror());
  R.create_handle = reinterpret_cast<int (*)(void**)>(dlsym(R.blas, "rocblas_create_handle"));
  R.destroy_handle = reinterpret_cast<int (*)(void*)>(dlsym(R.blas, "rocblas_destroy_handle"));
  R.set_stream = reinterpret_cast<int (*)(void*, hipStream_t)>(dlsym(R.blas, "rocblas_set_stream"));
  R.dtrsm = reinterpret_cast<decltype(R.dtrsm)>(dlsym(R.blas, "rocblas_dtrsm"));
  R.dgemm = reinterpret_cast<decltype(R.dgemm)>(dlsym(R.blas, "rocblas_dgemm"));
  R.dsyrk = reinterpret_cast<decltype(R.dsyrk)>(dlsym(R.blas, "rocblas_dsyrk"));
  R.dgemv = reinterpret_cast<decltype(R.dgemv)>(dlsym(R.blas, "rocblas_dgemv"));
  R.dtrsv = reinterpret_cast<decltype(R.dtrsv)>(dlsym(R.blas, "rocblas_dtrsv"));
  if (!R.create_handle || !R.destroy_handle || !R.set_stream || !R.dtrsm || !R.dgemm || !R.dsyrk || !R.dgemv || !R.dtrsv)
    return er::fail("rocBLAS symbols not found");
  if (R.create_handle(&R.handle) != 0 || R.set_stream(R.handle, stream) != 0) {
    R.handle = nullptr;
    return er::fail("rocblas_create_handle failed");
  }
  return 0;
}

}  // namespace

struct er_fopt_s {
  int device = 0, num = 0, res = 8, nper = 0;
  float length = 3.0f, ul = 0.375f;
  hipStream_t stream = nullptr;
  struct Frag {
    int n = 0;
    int* idx0 = nullptr;
    float *val = nullptr, *nval = nullptr, *p = nullptr, *nrm = nullptr;
    std::vector<int> h_idx0;
  };
  std::vector<Frag> frag;
  FragPtr* d_frags = nullptr;
  int n_pairs = 0, n_chunks = 0, n_groups = 0;
  std::vector<int> group_info;   // 4 per group: fragment i, fragment j, idx_[0] of the cell of p_i, of p_j
  double *d_diag = nullptr, *d_off = nullptr;
  size_t diag_cap = 0, off_cap = 0;
  // the system kept on the device for er_fopt_solve
  RocBlas roc;
  double *d_sys = nullptr, *d_rhs = nullptr;       // d_sys aliases d_JJ in the SLAC mode, own allocation in the non-rigid mode
  double* d_big = nullptr;
  size_t big_cap = 0;
  // block-sparse lower triangle of the non-rigid system (fragment blocks; er_fopt_factor_nonrigid when the dense matrix is too big)
  bool blocked = false;
  std::vector<long> blk_off;                       // [nb * nb]: offset (doubles) of block (row, col), row >= col, or -1
  std::vector<std::vector<int>> blk_rows;          // per block column: rows > column with a structurally non-zero block (ascending)
  long* d_blk_off = nullptr;
  long sys_n = 0;
  int* d_info = nullptr;
  int* d_ginfo = nullptr;
  long shift_index = -1;         // er_fopt_debug_shift_diagonal: added to one diagonal entry of every system before it is factored
  double shift_value = 0.0;
  bool factored = false;
  long n_corr = 0;
  int *d_first = nullptr, *d_second = nullptr;
  Chunk* d_chunks = nullptr;
  double *d_JJ = nullptr, *d_Jb = nullptr, *d_rot = nullptr, *d_ctr = nullptr;
  float* d_M = nullptr;
  size_t jj_cap = 0;
};

namespace {

void free_frag(er_fopt_s::Frag& f) {
  void* ptrs[] = {f.idx0, f.val, f.nval, f.p, f.nrm};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  f = er_fopt_s::Frag();
}

// PointCloud::GetCoordinate, PointCloud.h:92-176, on the host (once per point at load time; float32 like the reference)
bool get_coordinate(int res, float ul, const float* in6, int& idx0, float val[8], float nval[8], float nrm[3]) {
  float pt[6];
  memcpy(pt, in6, sizeof pt);
  const int corner[3] = {(int)floor(pt[0] / ul), (int)floor(pt[1] / ul), (int)floor(pt[2] / ul)};
  if (corner[0] < 0 || corner[0] >= res || corner[1] < 0 || corner[1] >= res || corner[2] < 0 || corner[2] >= res) return false;
  const float r[3] = {pt[0] / ul - corner[0], pt[1] / ul - corner[1], pt[2] / ul - corner[2]};
  idx0 = (corner[0] + corner[1] * (res + 1) + corner[2] * (res + 1) * (res + 1)) * 3;
  val[0] = (1 - r[0]) * (1 - r[1]) * (1 - r[2]);
  val[1] = (1 - r[0]) * (1 - r[1]) * (r[2]);
  val[2] = (1 - r[0]) * (r[1]) * (1 - r[2]);
  val[3] = (1 - r[0]) * (r[1]) * (r[2]);
  val[4] = (r[0]) * (1 - r[1]) * (1 - r[2]);
  val[5] = (r[0]) * (1 - r[1]) * (r[2]);
  val[6] = (r[0]) * (r[1]) * (1 - r[2]);
  val[7] = (r[0]) * (r[1]) * (r[2]);
  pt[3] /= ul;
  pt[4] /= ul;
  pt[5] /= ul;
  nval[0] = -pt[3] * (1 - r[1]) * (1 - r[2]) - pt[4] * (1 - r[0]) * (1 - r[2]) - pt[5] * (1 - r[0]) * (1 - r[1]);
  nval[1] = -pt[3] * (1 - r[1]) * (r[2]) - pt[4] * (1 - r[0]) * (r[2]) + pt[5] * (1 - r[0]) * (1 - r[1]);
  nval[2] = -pt[3] * (r[1]) * (1 - r[2]) + pt[4] * (1 - r[0]) * (1 - r[2]) - pt[5] * (1 - r[0]) * (r[1]);
  nval[3] = -pt[3] * (r[1]) * (r[2]) + pt[4] * (1 - r[0]) * (r[2]) + pt[5] * (1 - r[0]) * (r[1]);
  nval[4] = pt[3] * (1 - r[1]) * (1 - r[2]) - pt[4] * (r[0]) * (1 - r[2]) - pt[5] * (r[0]) * (1 - r[1]);
  nval[5] = pt[3] * (1 - r[1]) * (r[2]) - pt[4] * (r[0]) * (r[2]) + pt[5] * (r[0]) * (1 - r[1]);
  nval[6] = pt[3] * (r[1]) * (1 - r[2]) + pt[4] * (r[0]) * (1 - r[2]) - pt[5] * (r[0]) * (r[1]);
  nval[7] = pt[3] * (r[1]) * (r[2]) + pt[4] * (r[0]) * (r[2]) + pt[5] * (r[0]) * (r[1]);
  nrm[0] = pt[3];
  nrm[1] = pt[4];
  nrm[2] = pt[5];
  return true;
}

int upload_frag_table(er_fopt_t h) {
  std::vector<FragPtr> t((size_t)h->num);
  for (int f = 0; f < h->num; f++) t[(size_t)f] = FragPtr{h->frag[(size_t)f].idx0, h->frag[(size_t)f].val, h->frag[(size_t)f].p, h->frag[(size_t)f].nrm};
  ER_HIP_TRY(hipMemcpy(h->d_frags, t.data(), t.size() * sizeof(FragPtr), hipMemcpyHostToDevice));
  return 0;
}

int ensure_matrix(er_fopt_t h, size_t N) {
  if (h->d_sys == h->d_JJ) h->factored = false;                  // the caller is about to overwrite a SLAC factor kept in d_JJ
  if (N * N <= h->jj_cap) return 0;
  if (h->d_JJ) (void)hipFree(h->d_JJ);
  if (h->d_Jb) (void)hipFree(h->d_Jb);
  h->d_JJ = h->d_Jb = nullptr;
  h->jj_cap = 0;
  ER_HIP_TRY(hipMalloc((void**)&h->d_JJ, N * N * sizeof(double)));
  ER_HIP_TRY(hipMalloc((void**)&h->d_Jb, (N + 1) * sizeof(double)));
  h->jj_cap = N * N;
  return 0;
}

template <int MODE>
int assemble(er_fopt_t h, const double* pose_rot_t, double* JJ, double* Jb, double* score) {
  if (!JJ || !Jb || !score) return er::fail("er_fopt_assemble: NULL output");
  ER_HIP_TRY(hipSetDevice(h->device));
  const size_t N = MODE ? (size_t)(6 * h->num + h->nper) : (size_t)(6 * h->num);
  if (ensure_matrix(h, N)) return 1;
  ER_HIP_TRY(hipMemsetAsync(h->d_JJ, 0, N * N * sizeof(double), h->stream));
  ER_HIP_TRY(hipMemsetAsync(h->d_Jb, 0, (N + 1) * sizeof(double), h->stream));
  if (MODE) {
    if (!pose_rot_t) return er::fail("er_fopt_assemble_slac: pose_rot_t is NULL");
    ER_HIP_TRY(hipMemcpyAsync(h->d_rot, pose_rot_t, (size_t)h->num * 9 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  }
  if (h->n_chunks > 0) {
    const int blocks = (h->n_chunks * 64 + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_fopt_gram<MODE>, dim3(blocks), dim3(kBlock), 0, h->stream, h->d_chunks, h->n_chunks, h->d_frags, h->d_first, h->d_second,
                       h->d_rot, h->num, h->res, (int)N, h->d_JJ, h->d_Jb, h->d_Jb + N);
    ER_HIP_TRY(hipGetLastError());
  }
  ER_HIP_TRY(hipMemcpyAsync(JJ, h->d_JJ, N * N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  ER_HIP_TRY(hipMemcpyAsync(Jb, h->d_Jb, N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  ER_HIP_TRY(hipMemcpyAsync(score, h->d_Jb + N, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  if (MODE == 0)
    for (int k = 0; k < 6 && k < (int)N; k++) JJ[(size_t)k * N + k] += (double)h->n_pairs;     // mat_adder.Add( k, k, 1 ) per pair, OptApp.cpp:322-324
  return 0;
}

}  // namespace

extern "C" {

int er_fopt_create(int num, int resolution, float length, int device, er_fopt_t* out) {
  if (!out) return er::fail("er_fopt_create: out is NULL");
  *out = nullptr;
  if (num <= 0 || resolution <= 0 || !(length > 0.f)) return er::fail("er_fopt_create: bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return er::fail("er_fopt_create: no HIP device available (liber_hip has no CPU fallback)");
  if (device < 0 || device >= ndev) return er::fail("er_fopt_create: device %d out of range [0,%d)", device, ndev);
  ER_HIP_TRY(hipSetDevice(device));
  er_fopt_t h = new er_fopt_s();
  h->device = device;
  h->num = num;
  h->res = resolution;
  h->length = length;
  h->ul = length / resolution;                                  // PointCloud.cpp:10
  h->nper = (resolution + 1) * (resolution + 1) * (resolution + 1) * 3;
  h->frag.resize((size_t)num);
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&h->d_frags, (size_t)num * sizeof(FragPtr)) != hipSuccess ||
      hipMalloc((void**)&h->d_rot, (size_t)num * 9 * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&h->d_ctr, (size_t)h->nper * sizeof(double)) != hipSuccess || hipMalloc((void**)&h->d_M, 16 * sizeof(float)) != hipSuccess) {
    const hipError_t e = hipGetLastError();
    er_fopt_destroy(h);
    return er::fail("er_fopt_create: allocation failed: %s", hipGetErrorString(e));
  }
  if (upload_frag_table(h)) {
    er_fopt_destroy(h);
    return 1;
  }
  *out = h;
  return 0;
}

int er_fopt_destroy(er_fopt_t h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto& f : h->frag) free_frag(f);
  if (h->roc.handle && h->roc.destroy_handle) (void)h->roc.destroy_handle(h->roc.handle);
  void* ptrs[] = {h->d_frags, h->d_first, h->d_second, h->d_chunks, h->d_JJ, h->d_Jb, h->d_rot, h->d_ctr, h->d_M, h->d_diag, h->d_off,
                  h->d_big, h->d_rhs, h->d_info, h->d_ginfo, h->d_blk_off};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

int er_fopt_set_cloud(er_fopt_t h, int frag, const float* xyz, const float* nrm, int n, int* first_out_of_bound) {
  if (!h || frag < 0 || frag >= h->num || n < 0 || (n > 0 && (!xyz || !nrm))) return er::fail("er_fopt_set_cloud: bad arguments");
  ER_HIP_TRY(hipSetDevice(h->device));
  if (first_out_of_bound) *first_out_of_bound = -1;
  std::vector<int> idx0((size_t)n);
  std::vector<float> val((size_t)n * 8), nval((size_t)n * 8), p((size_t)n * 3), nn((size_t)n * 3);
  int m = n;
  for (int k = 0; k < n; k++) {
    const float x[6] = {xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2], nrm[3 * k], nrm[3 * k + 1], nrm[3 * k + 2]};
    memcpy(&p[(size_t)k * 3], x, 3 * sizeof(float));
    if (!get_coordinate(h->res, h->ul, x, idx0[(size_t)k], &val[(size_t)k * 8], &nval[(size_t)k * 8], &nn[(size_t)k * 3])) {
      if (first_out_of_bound) *first_out_of_bound = k;          // "Error!! Point out of bound!!" -- loading stops here (PointCloud.cpp:57-60)
      m = k;
      break;
    }
  }
  er_fopt_s::Frag& f = h->frag[(size_t)frag];
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  free_frag(f);
  // correspondence lists index into the clouds: a new cloud invalidates them (set them again) and any factored system
  h->n_pairs = h->n_chunks = h->n_groups = 0;
  h->group_info.clear();
  h->factored = false;
  const size_t mm = (size_t)std::max(m, 1);
  const bool ok = hipMalloc((void**)&f.idx0, mm * sizeof(int)) == hipSuccess && hipMalloc((void**)&f.val, mm * 8 * sizeof(float)) == hipSuccess &&
                  hipMalloc((void**)&f.nval, mm * 8 * sizeof(float)) == hipSuccess && hipMalloc((void**)&f.p, mm * 3 * sizeof(float)) == hipSuccess &&
                  hipMalloc((void**)&f.nrm, mm * 3 * sizeof(float)) == hipSuccess &&
                  (m == 0 || (hipMemcpy(f.idx0, idx0.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
                              hipMemcpy(f.val, val.data(), (size_t)m * 8 * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
                              hipMemcpy(f.nval, nval.data(), (size_t)m * 8 * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
                              hipMemcpy(f.p, p.data(), (size_t)m * 3 * sizeof(float), hipMemcpyHostToDevice) == hipSuccess &&
                              hipMemcpy(f.nrm, nn.data(), (size_t)m * 3 * sizeof(float), hipMemcpyHostToDevice) == hipSuccess));
  if (!ok) {                                                     // leave an empty fragment behind, never a half-built one
    const hipError_t e = hipGetLastError();
    free_frag(f);
    (void)upload_frag_table(h);
    return er::fail("er_fopt_set_cloud: fragment %d (%d points): %s", frag, m, hipGetErrorString(e));
  }
  f.n = m;
  f.h_idx0.assign(idx0.begin(), idx0.begin() + m);
  return upload_frag_table(h);
}

int er_fopt_cloud_size(er_fopt_t h, int frag) { return (h && frag >= 0 && frag < h->num) ? h->frag[(size_t)frag].n : -1; }

int er_fopt_get_points(er_fopt_t h, int frag, int* idx0, float* val, float* nval, float* p, float* nrm) {
  if (!h || frag < 0 || frag >= h->num) return er::fail("er_fopt_get_points: bad arguments");
  ER_HIP_TRY(hipSetDevice(h->device));
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  const er_fopt_s::Frag& f = h->frag[(size_t)frag];
  const size_t m = (size_t)f.n;
  if (m == 0) return 0;
  if (idx0) ER_HIP_TRY(hipMemcpy(idx0, f.idx0, m * sizeof(int), hipMemcpyDeviceToHost));
  if (val) ER_HIP_TRY(hipMemcpy(val, f.val, m * 8 * sizeof(float), hipMemcpyDeviceToHost));
  if (nval) ER_HIP_TRY(hipMemcpy(nval, f.nval, m * 8 * sizeof(float), hipMemcpyDeviceToHost));
  if (p) ER_HIP_TRY(hipMemcpy(p, f.p, m * 3 * sizeof(float), hipMemcpyDeviceToHost));
  if (nrm) ER_HIP_TRY(hipMemcpy(nrm, f.nrm, m * 3 * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int er_fopt_update_pose(er_fopt_t h, int frag, const float M[16]) {
  if (!h || frag < 0 || frag >= h->num || !M) return er::fail("er_fopt_update_pose: bad arguments");
  ER_HIP_TRY(hipSetDevice(h->device));
  er_fopt_s::Frag& f = h->frag[(size_t)frag];
  if (f.n == 0) return 0;
  ER_HIP_TRY(hipStreamSynchronize(h->stream));                   // d_M is reused call after call
  ER_HIP_TRY(hipMemcpyAsync(h->d_M, M, 16 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_fopt_update_pose, dim3((f.n + kBlock - 1) / kBlock), dim3(kBlock), 0, h->stream, f.p, f.nrm, f.n, h->d_M);
  ER_HIP_TRY(hipGetLastError());
  return 0;
}

static int update_from_ctr(er_fopt_t h, int frag, const double* ctr_slice, int normals_only) {
  if (!h || frag < 0 || frag >= h->num || !ctr_slice) return er::fail("er_fopt_update_point_pn / update_normals: bad arguments");
  ER_HIP_TRY(hipSetDevice(h->device));
  er_fopt_s::Frag& f = h->frag[(size_t)frag];
  if (f.n == 0) return 0;
  ER_HIP_TRY(hipStreamSynchronize(h->stream));                   // d_ctr is reused call after call
  ER_HIP_TRY(hipMemcpyAsync(h->d_ctr, ctr_slice, (size_t)h->nper * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_fopt_update_pn, dim3((f.n + kBlock - 1) / kBlock), dim3(kBlock), 0, h->stream, f.idx0, f.val, f.nval, f.p, f.nrm, f.n,
                     h->d_ctr, h->res, normals_only);
  ER_HIP_TRY(hipGetLastError());
  return 0;
}

int er_fopt_update_point_pn(er_fopt_t h, int frag, const double* ctr_slice) { return update_from_ctr(h, frag, ctr_slice, 0); }
int er_fopt_update_normals(er_fopt_t h, int frag, const double* ctr_slice) { return update_from_ctr(h, frag, ctr_slice, 1); }

int er_fopt_set_correspondences(er_fopt_t h, int n_pairs, const int* frag_i, const int* frag_j, const int* const* pairs, const int* counts) {
  if (!h || n_pairs < 0 || (n_pairs > 0 && (!frag_i || !frag_j || !pairs || !counts))) return er::fail("er_fopt_set_correspondences: bad arguments");
  ER_HIP_TRY(hipSetDevice(h->device));
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  std::vector<int> first, second, ginfo;
  std::vector<Chunk> chunks;
  for (int l = 0; l < n_pairs; l++) {
    const int i = frag_i[l], j = frag_j[l], m = counts[l];
    if (i < 0 || i >= h->num || j < 0 || j >= h->num || m < 0 || (m > 0 && !pairs[l])) return er::fail("er_fopt_set_correspondences: bad pair %d", l);
    const std::vector<int>& ci = h->frag[(size_t)i].h_idx0;
    const std::vector<int>& cj = h->frag[(size_t)j].h_idx0;
    std::vector<long long> key((size_t)m);
    std::vector<int> order((size_t)m);
    for (int k = 0; k < m; k++) {
      const int a = pairs[l][2 * k], b = pairs[l][2 * k + 1];
      if (a < 0 || a >= (int)ci.size() || b < 0 || b >= (int)cj.size())
        return er::fail("er_fopt_set_correspondences: pair %d row %d (%d, %d) out of range (%zu, %zu points)", l, k, a, b, ci.size(), cj.size());
      key[(size_t)k] = (long long)ci[(size_t)a] * (1LL << 32) + cj[(size_t)b];
      order[(size_t)k] = k;
    }
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return key[(size_t)x] < key[(size_t)y]; });
    int g0 = 0;
    while (g0 < m) {
      int g1 = g0;
      while (g1 < m && key[(size_t)order[(size_t)g1]] == key[(size_t)order[(size_t)g0]]) g1++;
      const int a0 = pairs[l][2 * order[(size_t)g0]], b0 = pairs[l][2 * order[(size_t)g0] + 1];
      const int gid = (int)ginfo.size() / 4;
      ginfo.insert(ginfo.end(), {i, j, ci[(size_t)a0], cj[(size_t)b0]});
      for (int s = g0; s < g1; s += kChunkMax)
        chunks.push_back(Chunk{i, j, ci[(size_t)a0], cj[(size_t)b0], (int)first.size() + (s - g0), std::min(kChunkMax, g1 - s), gid});
      for (int s = g0; s < g1; s++) {
        first.push_back(pairs[l][2 * order[(size_t)s]]);
        second.push_back(pairs[l][2 * order[(size_t)s] + 1]);
      }
      g0 = g1;
    }
  }
  void* old[] = {h->d_first, h->d_second, h->d_chunks};
  for (void* p : old)
    if (p) (void)hipFree(p);
  h->d_first = h->d_second = nullptr;
  h->d_chunks = nullptr;
  h->n_pairs = h->n_chunks = h->n_groups = 0;                    // an upload failure below leaves "no correspondences", not dangling lists
  h->n_corr = 0;
  h->group_info.clear();
  if (h->d_ginfo) {
    (void)hipFree(h->d_ginfo);
    h->d_ginfo = nullptr;
  }
  h->factored = false;
  if (!first.empty()) {
    ER_HIP_TRY(hipMalloc((void**)&h->d_first, first.size() * sizeof(int)));
    ER_HIP_TRY(hipMalloc((void**)&h->d_second, second.size() * sizeof(int)));
    ER_HIP_TRY(hipMalloc((void**)&h->d_chunks, chunks.size() * sizeof(Chunk)));
    ER_HIP_TRY(hipMemcpy(h->d_first, first.data(), first.size() * sizeof(int), hipMemcpyHostToDevice));
    ER_HIP_TRY(hipMemcpy(h->d_second, second.data(), second.size() * sizeof(int), hipMemcpyHostToDevice));
    ER_HIP_TRY(hipMemcpy(h->d_chunks, chunks.data(), chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice));
  }
  h->n_pairs = n_pairs;
  h->n_chunks = (int)chunks.size();
  h->n_groups = (int)ginfo.size() / 4;
  h->group_info.swap(ginfo);
  h->n_corr = (long)first.size();
  return 0;
}

// ---- the same with the lists already in HBM (round 5; VERDICT round 4: "a device-resident hand-off between the two paths' consumers") --------------
// key = list << 40 | idx_[0] of p_i << 20 | idx_[0] of p_j: ONE stable radix sort over all lists reproduces the host path's order -- lists in the given
// order, groups by ascending (cell, cell), rows of a group in list order -- so d_first / d_second, the chunks and with them every float64 sum of the
// assembly come out bit-identical to er_fopt_set_correspondences on the downloaded lists (tests/test_fopt_gpu.py).
namespace {
__global__ __launch_bounds__(kBlock) void k_corr_keys(const int* const* __restrict__ lists, const long* __restrict__ off, const int* __restrict__ fi,
                                                      const int* __restrict__ fj, const FragPtr* __restrict__ frags, const int* __restrict__ frag_n,
                                                      unsigned long long* __restrict__ keys, unsigned* __restrict__ vals, int* __restrict__ a_out,
                                                      int* __restrict__ b_out, int* __restrict__ bad) {
  const int l = blockIdx.y;
  const long o = off[l];
  const int count = (int)(off[l + 1] - o);
  const int i = fi[l], j = fj[l];
  const int* __restrict__ rows = lists[l];
  for (int k = blockIdx.x * kBlock + threadIdx.x; k < count; k += gridDim.x * kBlock) {
    const int a = rows[2 * k], b = rows[2 * k + 1];
    const bool ok = a >= 0 && a < frag_n[i] && b >= 0 && b < frag_n[j];
    if (!ok && atomicCAS(&bad[0], 0, 1) == 0) {                 // the first offender that gets here is reported
      bad[1] = l;
      bad[2] = k;
      bad[3] = a;
      bad[4] = b;
    }
    const unsigned long long ci = ok ? (unsigned)frags[i].idx0[a] : 0u, cj = ok ? (unsigned)frags[j].idx0[b] : 0u;
    keys[o + k] = ((unsigned long long)l << 40) | (ci << 20) | cj;
    vals[o + k] = (unsigned)(o + k);
    a_out[o + k] = a;
    b_out[o + k] = b;
  }
}
__global__ __launch_bounds__(kBlock) void k_corr_gather(const unsigned* __restrict__ order, const int* __restrict__ a_in, const int* __restrict__ b_in, long n,
                                                        int* __restrict__ first, int* __restrict__ second) {
  const long s = (long)blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  const unsigned g = order[s];
  first[s] = a_in[g];
  second[s] = b_in[g];
}
}  // namespace

int er_fopt_set_correspondences_dev(er_fopt_t h, int n_pairs, const int* frag_i, const int* frag_j, const int* const* pairs_dev, const int* counts) {
  if (!h || n_pairs < 0 || (n_pairs > 0 && (!frag_i || !frag_j || !pairs_dev || !counts))) return er::fail("er_fopt_set_correspondences_dev: bad arguments");
  if (n_pairs >= (1 << 23)) return er::fail("er_fopt_set_correspondences_dev: %d lists exceed the 2^23 the sort key holds", n_pairs);
  if ((long)(h->res + 1) * (h->res + 1) * (h->res + 1) * 3 >= (1L << 20)) return er::fail("er_fopt_set_correspondences_dev: the lattice is too fine for the sort key");
  ER_HIP_TRY(hipSetDevice(h->device));
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  std::vector<long> off((size_t)n_pairs + 1, 0);
  std::vector<int> fn((size_t)h->num);
  for (int q = 0; q < h->num; q++) fn[(size_t)q] = h->frag[(size_t)q].n;
  for (int l = 0; l < n_pairs; l++) {
    const int i = frag_i[l], j = frag_j[l], m = counts[l];
    if (i < 0 || i >= h->num || j < 0 || j >= h->num || m < 0 || (m > 0 && !pairs_dev[l])) return er::fail("er_fopt_set_correspondences_dev: bad pair %d", l);
    if (m > 0 && (fn[(size_t)i] == 0 || fn[(size_t)j] == 0 || !h->d_frags))      // (the key kernel reads the fragments' idx_[0] arrays)
      return er::fail("er_fopt_set_correspondences_dev: pair %d refers to a fragment without a cloud (er_fopt_set_cloud first)", l);
    off[(size_t)l + 1] = off[(size_t)l] + m;
  }
  const long N = off[(size_t)n_pairs];
  if (N >= (1L << 31)) return er::fail("er_fopt_set_correspondences_dev: %ld correspondences exceed 2^31 - 1", N);
  // the old lists go first (as in the host path: a failure below leaves "no correspondences", not dangling lists)
  void* old[] = {h->d_first, h->d_second, h->d_chunks, h->d_ginfo};
  for (void* p : old)
    if (p) (void)hipFree(p);
  h->d_first = h->d_second = nullptr;
  h->d_chunks = nullptr;
  h->d_ginfo = nullptr;
  h->n_pairs = h->n_chunks = h->n_groups = 0;
  h->n_corr = 0;
  h->group_info.clear();
  h->factored = false;
  if (N == 0) {
    h->n_pairs = n_pairs;
    return 0;
  }
  int rc = 0;
  const int** d_lists = nullptr;
  long* d_off = nullptr;
  int *d_fi = nullptr, *d_fj = nullptr, *d_fn = nullptr, *d_a = nullptr, *d_b = nullptr, *d_bad = nullptr, *d_cnt = nullptr, *d_runs = nullptr, *d_first = nullptr,
      *d_second = nullptr;
  unsigned long long *d_k0 = nullptr, *d_k1 = nullptr, *d_uni = nullptr;
  unsigned *d_v0 = nullptr, *d_v1 = nullptr;
  void* d_tmp = nullptr;
  std::vector<unsigned long long> uni;
  std::vector<int> cnt, ginfo;
  std::vector<Chunk> chunks;
  int bad[5] = {0, 0, 0, 0, 0}, runs = 0;
  size_t need_sort = 0, need_rle = 0;
  int lbits = 1;
  while ((1L << lbits) < (long)n_pairs) lbits++;
#define ER_D(expr)                                                                                      \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) {                                                                             \
      rc = er::fail("er_fopt_set_correspondences_dev: %s failed: %s", #expr, hipGetErrorString(e_));    \
      goto done;                                                                                        \
    }                                                                                                   \
  } while (0)
  ER_D(hipMalloc((void**)&d_lists, (size_t)n_pairs * sizeof(int*)));
  ER_D(hipMalloc((void**)&d_off, ((size_t)n_pairs + 1) * sizeof(long)));
  ER_D(hipMalloc((void**)&d_fi, (size_t)n_pairs * sizeof(int)));
  ER_D(hipMalloc((void**)&d_fj, (size_t)n_pairs * sizeof(int)));
  ER_D(hipMalloc((void**)&d_fn, (size_t)h->num * sizeof(int)));
  ER_D(hipMalloc((void**)&d_bad, 5 * sizeof(int)));
  ER_D(hipMalloc((void**)&d_runs, sizeof(int)));
  ER_D(hipMalloc((void**)&d_k0, (size_t)N * 8));
  ER_D(hipMalloc((void**)&d_k1, (size_t)N * 8));
  ER_D(hipMalloc((void**)&d_uni, (size_t)N * 8));
  ER_D(hipMalloc((void**)&d_v0, (size_t)N * 4));
  ER_D(hipMalloc((void**)&d_v1, (size_t)N * 4));
  ER_D(hipMalloc((void**)&d_a, (size_t)N * 4));
  ER_D(hipMalloc((void**)&d_b, (size_t)N * 4));
  ER_D(hipMalloc((void**)&d_cnt, (size_t)N * 4));
  ER_D(hipMalloc((void**)&d_first, (size_t)N * 4));
  ER_D(hipMalloc((void**)&d_second, (size_t)N * 4));
  ER_D(hipMemcpyAsync(d_lists, pairs_dev, (size_t)n_pairs * sizeof(int*), hipMemcpyHostToDevice, h->stream));
  ER_D(hipMemcpyAsync(d_off, off.data(), ((size_t)n_pairs + 1) * sizeof(long), hipMemcpyHostToDevice, h->stream));
  ER_D(hipMemcpyAsync(d_fi, frag_i, (size_t)n_pairs * sizeof(int), hipMemcpyHostToDevice, h->stream));
  ER_D(hipMemcpyAsync(d_fj, frag_j, (size_t)n_pairs * sizeof(int), hipMemcpyHostToDevice, h->stream));
  ER_D(hipMemcpyAsync(d_fn, fn.data(), (size_t)h->num * sizeof(int), hipMemcpyHostToDevice, h->stream));
  ER_D(hipMemsetAsync(d_bad, 0, 5 * sizeof(int), h->stream));
  {
    int mx = 1;
    for (int l = 0; l < n_pairs; l++) mx = std::max(mx, counts[l]);
    hipLaunchKernelGGL(k_corr_keys, dim3(std::min((mx + kBlock - 1) / kBlock, 1024), n_pairs), dim3(kBlock), 0, h->stream, d_lists, d_off, d_fi, d_fj, h->d_frags, d_fn,
                       d_k0, d_v0, d_a, d_b, d_bad);
  }
  ER_D(hipGetLastError());
  ER_D(hipcub::DeviceRadixSort::SortPairs(nullptr, need_sort, d_k0, d_k1, d_v0, d_v1, (int)N, 0, 40 + lbits, h->stream));
  ER_D(hipcub::DeviceRunLengthEncode::Encode(nullptr, need_rle, d_k1, d_uni, d_cnt, d_runs, (int)N, h->stream));
  {
    size_t need = std::max(need_sort, need_rle);
    ER_D(hipMalloc(&d_tmp, need));
    size_t t = need;
    ER_D(hipcub::DeviceRadixSort::SortPairs(d_tmp, t, d_k0, d_k1, d_v0, d_v1, (int)N, 0, 40 + lbits, h->stream));
    t = need;
    ER_D(hipcub::DeviceRunLengthEncode::Encode(d_tmp, t, d_k1, d_uni, d_cnt, d_runs, (int)N, h->stream));
  }
  hipLaunchKernelGGL(k_corr_gather, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, d_v1, d_a, d_b, N, d_first, d_second);
  ER_D(hipGetLastError());
  ER_D(hipMemcpyAsync(bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, h->stream));
  ER_D(hipMemcpyAsync(&runs, d_runs, sizeof runs, hipMemcpyDeviceToHost, h->stream));
  ER_D(hipStreamSynchronize(h->stream));
  if (bad[0]) {
    rc = er::fail("er_fopt_set_correspondences_dev: pair %d row %d (%d, %d) out of range (%d, %d points)", bad[1], bad[2], bad[3], bad[4],
                  fn[(size_t)frag_i[bad[1]]], fn[(size_t)frag_j[bad[1]]]);
    goto done;
  }
  uni.resize((size_t)runs);
  cnt.resize((size_t)runs);
  ER_D(hipMemcpy(uni.data(), d_uni, (size_t)runs * 8, hipMemcpyDeviceToHost));
  ER_D(hipMemcpy(cnt.data(), d_cnt, (size_t)runs * 4, hipMemcpyDeviceToHost));
  {
    long start = 0;
    for (int r = 0; r < runs; r++) {                            // the group table: the only part of the lists' structure that visits the host
      const unsigned long long key = uni[(size_t)r];
      const int l = (int)(key >> 40), ci = (int)((key >> 20) & 0xfffffu), cj = (int)(key & 0xfffffu), m = cnt[(size_t)r];
      const int i = frag_i[l], j = frag_j[l];
      ginfo.insert(ginfo.end(), {i, j, ci, cj});
      for (int s0 = 0; s0 < m; s0 += kChunkMax) chunks.push_back(Chunk{i, j, ci, cj, (int)(start + s0), std::min(kChunkMax, m - s0), r});
      start += m;
    }
  }
  ER_D(hipMalloc((void**)&h->d_chunks, chunks.size() * sizeof(Chunk)));
  ER_D(hipMemcpy(h->d_chunks, chunks.data(), chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice));
  h->d_first = d_first;
  h->d_second = d_second;
  d_first = d_second = nullptr;                                 // (now owned by the handle)
  h->n_pairs = n_pairs;
  h->n_chunks = (int)chunks.size();
  h->n_groups = runs;
  h->group_info.swap(ginfo);
  h->n_corr = N;
#undef ER_D
done:
  {
    void* tmp[] = {(void*)d_lists, d_off, d_fi, d_fj, d_fn, d_a, d_b, d_bad, d_cnt, d_runs, d_first, d_second, d_k0, d_k1, d_uni, d_v0, d_v1, d_tmp};
    for (void* p : tmp)
      if (p) (void)hipFree(p);
  }
  return rc;
}

int er_fopt_group_count(er_fopt_t h) { return h ? h->n_groups : -1; }

int er_fopt_group_info(er_fopt_t h, int* info4) {
  if (!h || !info4) return er::fail("er_fopt_group_info: bad arguments");
  if (!h->group_info.empty()) memcpy(info4, h->group_info.data(), h->group_info.size() * sizeof(int));
  return 0;
}

int er_fopt_assemble_rigid(er_fopt_t h, double* JJ, double* Jb, double* score) {
  if (!h) return er::fail("er_fopt_assemble_rigid: NULL handle");
  return assemble<0>(h, nullptr, JJ, Jb, score);
}

int er_fopt_assemble_slac(er_fopt_t h, const double* pose_rot_t, double* JJ, double* Jb, double* score) {
  if (!h) return er::fail("er_fopt_assemble_slac: NULL handle");
  return assemble<1>(h, pose_rot_t, JJ, Jb, score);
}

int er_fopt_assemble_nonrigid(er_fopt_t h, double weight, double* diag, double* offdiag) {
  if (!h || !diag || (h->n_groups > 0 && !offdiag)) return er::fail("er_fopt_assemble_nonrigid: bad arguments");
  ER_HIP_TRY(hipSetDevice(h->device));
  const size_t nv = (size_t)(h->res + 1) * (h->res + 1) * (h->res + 1);
  const size_t nd = (size_t)h->num * nv * 576, no = (size_t)std::max(h->n_groups, 1) * 576;
  if (nd > h->diag_cap) {
    if (h->d_diag) (void)hipFree(h->d_diag);
    h->d_diag = nullptr;
    h->diag_cap = 0;
    ER_HIP_TRY(hipMalloc((void**)&h->d_diag, nd * sizeof(double)));
    h->diag_cap = nd;
  }
  if (no > h->off_cap) {
    if (h->d_off) (void)hipFree(h->d_off);
    h->d_off = nullptr;
    h->off_cap = 0;
    ER_HIP_TRY(hipMalloc((void**)&h->d_off, no * sizeof(double)));
    h->off_cap = no;
  }
  ER_HIP_TRY(hipMemsetAsync(h->d_diag, 0, nd * sizeof(double), h->stream));
  ER_HIP_TRY(hipMemsetAsync(h->d_off, 0, no * sizeof(double), h->stream));
  ER_HIP_TRY(hipMemcpyAsync(h->d_rot, &weight, sizeof(double), hipMemcpyHostToDevice, h->stream));     // MODE 2 reads the weight from rot_t[0]
  if (h->n_chunks > 0) {
    const int blocks = (h->n_chunks * 64 + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_fopt_gram<2>, dim3(blocks), dim3(kBlock), 0, h->stream, h->d_chunks, h->n_chunks, h->d_frags, h->d_first, h->d_second,
                       h->d_rot, h->num, h->res, 0, h->d_diag, h->d_off, (double*)nullptr);
    ER_HIP_TRY(hipGetLastError());
  }
  ER_HIP_TRY(hipMemcpyAsync(diag, h->d_diag, nd * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (h->n_groups > 0) ER_HIP_TRY(hipMemcpyAsync(offdiag, h->d_off, (size_t)h->n_groups * 576 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

// ---- on-device solve -------------------------------------------------------------------------------------------------
constexpr int kNotPositiveDefinite = 2;

// Recursive blocked Cholesky, lower triangle of a column-major n x n matrix in place:  A = [A11 . ; A21 A22] -> factor A11 (recursively,
// down to k_potrf_block on 64 x 64 diagonal blocks), L21 = A21 L11^-T (rocblas_dtrsm), A22 -= L21 L21^T (rocblas_dsyrk), factor A22.  The
// halving keeps the level-3 updates large (compute-bound on the FP64 matrix cores) at every size from the 2 211 unknowns of the test scenes to
// the 109 350 of 50 fragments.  *h->d_info = 0, or the 1-based index of the first non-positive pivot (the rest of the matrix is then
// meaningless).  Replaces rocsolver_dpotrf (round 3): under multi-process load that routine reported non-positive pivots near the end of
// matrices that ARE positive definite (3-6 of 120-160 concurrent runs in round 2, the same assembled matrix factored fine on the host), and
// round 2 could only retry around it.
static int potrf_rec(er_fopt_t h, double* A, long n, long lda, long j_base) {
  if (n <= kPotrfNb) {
    hipLaunchKernelGGL(k_potrf_block, dim3(1), dim3(256), 0, h->stream, A, lda, (int)n, (int)j_base, h->d_info);
    return 0;
  }
  const double one = 1.0, minus = -1.0;
  const long n1 = ((n / 2 + kPotrfNb - 1) / kPotrfNb) * kPotrfNb, n2 = n - n1;
  if (potrf_rec(h, A, n1, lda, j_base)) return 1;
  if (h->roc.dtrsm(h->roc.handle, kSideRight, kFillLower, kOpT, kDiagNonUnit, (int)n2, (int)n1, &one, A, (int)lda, A + n1, (int)lda) != 0)
    return er::fail("rocblas_dtrsm failed (Cholesky, columns %ld..%ld)", j_base, j_base + n1);
  double* A22 = A + (size_t)n1 * lda + n1;
  if (h->roc.dsyrk(h->roc.handle, kFillLower, kOpN, (int)n2, (int)n1, &minus, A + n1, (int)lda, &one, A22, (int)lda) != 0)
    return er::fail("rocblas_dsyrk failed (Cholesky, columns %ld..%ld)", j_base, j_base + n1);
  return potrf_rec(h, A22, n2, lda, j_base + n1);
}

static int potrf_lower(er_fopt_t h, double* A, long n, long lda) {
  ER_HIP_TRY(hipMemsetAsync(h->d_info, 0, sizeof(int), h->stream));
  if (potrf_rec(h, A, n, lda, 0)) return 1;
  ER_HIP_TRY(hipGetLastError());
  return 0;
}

static int factor_common(er_fopt_t h, double* A, long n) {
  if (rocblas_load(h->roc, h->stream)) return 1;
  if (!h->d_info) ER_HIP_TRY(hipMalloc((void**)&h->d_info, sizeof(int)));
  if (h->d_rhs) {
    (void)hipFree(h->d_rhs);
    h->d_rhs = nullptr;
  }
  if (n > 2147483647L) return er::fail("system too large for the 32-bit rocBLAS interface (%ld unknowns)", n);
  ER_HIP_TRY(hipMalloc((void**)&h->d_rhs, (size_t)n * sizeof(double)));
  // ER_FOPT_DIAG=1 (debugging aid): keep a host copy of the assembled matrix and, if potrf_lower reports a non-positive pivot,
  // factor that copy on the host -- tells a wrong matrix (assembly) from a wrong factorisation.
  if (h->shift_index >= 0 && h->shift_index < n) {
    hipLaunchKernelGGL(k_fopt_add_diag, dim3(1), dim3(64), 0, h->stream, A, n, h->shift_index, 1, h->shift_value);
    ER_HIP_TRY(hipGetLastError());
  }
  std::vector<double> diag_copy;
  if (getenv("ER_FOPT_DIAG")) {
    diag_copy.resize((size_t)n * n);
    ER_HIP_TRY(hipMemcpyAsync(diag_copy.data(), A, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    ER_HIP_TRY(hipStreamSynchronize(h->stream));
  }
  if (potrf_lower(h, A, n, n)) return 1;
  int info = 0;
  ER_HIP_TRY(hipMemcpyAsync(&info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  if (info != 0 && !diag_copy.empty()) {
    // column-major lower triangle == the row-major upper triangle that was assembled: a(i, j) for j >= i at [i * n + j]
    std::vector<double>& M = diag_copy;
    long bad = 0;
    for (long j = 0; j < n && !bad; j++) {                   // plain right-looking Cholesky on the upper triangle (U^T U)
      double d = M[(size_t)j * n + j];
      for (long k = 0; k < j; k++) d -= M[(size_t)k * n + j] * M[(size_t)k * n + j];
      if (!(d > 0.0)) { bad = j + 1; break; }
      d = sqrt(d);
      M[(size_t)j * n + j] = d;
      for (long c = j + 1; c < n; c++) {
        double v = M[(size_t)j * n + c];
        for (long k = 0; k < j; k++) v -= M[(size_t)k * n + j] * M[(size_t)k * n + c];
        M[(size_t)j * n + c] = v / d;
      }
    }
    er::fail("the assembled system is not positive definite (Cholesky pivot %d); host Cholesky of the SAME assembled matrix: %s (pivot %ld)", info,
             bad ? "ALSO not positive definite -> the matrix is wrong" : "positive definite -> the factorisation is wrong", bad);
    return kNotPositiveDefinite;
  }
  if (info != 0) {
    er::fail("the assembled system is not positive definite (Cholesky pivot %d)", info);
    return kNotPositiveDefinite;
  }
  h->d_sys = A;
  h->sys_n = n;
  h->blocked = false;
  h->factored = true;
  return 0;
}

static int factor_slac_once(er_fopt_t h, const double* pose_rot_t, double default_weight, double* dataJb_host, double* score);

int er_fopt_factor_slac(er_fopt_t h, const double* pose_rot_t, double default_weight, double* dataJb_host, double* score) {
  return factor_slac_once(h, pose_rot_t, default_weight, dataJb_host, score) ? 1 : 0;
}

static int factor_slac_once(er_fopt_t h, const double* pose_rot_t, double default_weight, double* dataJb_host, double* score) {
  if (!h || !pose_rot_t) return er::fail("er_fopt_factor_slac: bad arguments");
  ER_HIP_TRY(hipSetDevice(h->device));
  h->factored = false;
  const size_t N = (size_t)(6 * h->num + h->nper);
  if (ensure_matrix(h, N)) return 1;
  ER_HIP_TRY(hipMemsetAsync(h->d_JJ, 0, N * N * sizeof(double), h->stream));
  ER_HIP_TRY(hipMemsetAsync(h->d_Jb, 0, (N + 1) * sizeof(double), h->stream));
  ER_HIP_TRY(hipMemcpyAsync(h->d_rot, pose_rot_t, (size_t)h->num * 9 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (h->n_chunks > 0) {
    hipLaunchKernelGGL(k_fopt_gram<1>, dim3((h->n_chunks * 64 + kBlock - 1) / kBlock), dim3(kBlock), 0, h->stream, h->d_chunks, h->n_chunks,
                       h->d_frags, h->d_first, h->d_second, h->d_rot, h->num, h->res, (int)N, h->d_JJ, h->d_Jb, h->d_Jb + N);
  }
  // + default_weight * ( lattice Laplacian + anchor ) + the gauge "+1" on the first six unknowns (OptApp.cpp:452-464, 839-843)
  const int nv = h->nper / 3;
  const long L0 = 6L * h->num;
  hipLaunchKernelGGL(k_fopt_add_laplacian, dim3((nv * 6 + kBlock - 1) / kBlock), dim3(kBlock), 0, h->stream, h->d_JJ, (long)N, L0, h->res, default_weight);
  const long anchor = L0 + ((long)(h->res / 2) + (long)(h->res / 2) * (h->res + 1)) * 3;
  hipLaunchKernelGGL(k_fopt_add_diag, dim3(1), dim3(64), 0, h->stream, h->d_JJ, (long)N, anchor, 3, default_weight);
  hipLaunchKernelGGL(k_fopt_add_diag, dim3(1), dim3(64), 0, h->stream, h->d_JJ, (long)N, 0L, 6, 1.0);
  ER_HIP_TRY(hipGetLastError());
  if (dataJb_host) ER_HIP_TRY(hipMemcpyAsync(dataJb_host, h->d_Jb, N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (score) ER_HIP_TRY(hipMemcpyAsync(score, h->d_Jb + N, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  return factor_common(h, h->d_JJ, (long)N);
}

// ---- block-sparse Cholesky of the non-rigid system --------------------------------------------------------------------
// The reference factors thisAA with CHOLMOD's supernodal sparse Cholesky on the host (OptApp.cpp:155-211).  Here the
// matrix is kept as the lower triangle of FRAGMENT blocks (nper x nper = 2187 x 2187 at resolution 8, column-major): block
// (i, j) exists iff fragments i and j share a correspondence list or it fills in during the elimination (symbolic pass on
// the host over the fragment graph, natural order); the numeric factorisation is right-looking over those dense blocks --
// potrf_lower (below) on the diagonal block, rocBLAS trsm down the column, syrk / gemm into the trailing blocks -- and a solve is
// a block forward / backward substitution (trsv + gemv).  A 100-fragment scene whose pairs link neighbours needs a few
// hundred 38 MB blocks; even the complete graph of 100 fragments (5050 blocks, 193 GB) fits the 288 GB of one MI355X, where
// the dense square of the same system (383 GB) does not.
static int block_symbolic(er_fopt_t h) {
  const int nb = h->num;
  std::vector<std::vector<char>> nz((size_t)nb, std::vector<char>((size_t)nb, 0));
  for (int i = 0; i < nb; i++) nz[(size_t)i][(size_t)i] = 1;
  for (int g = 0; g < h->n_groups; g++) {
    const int a = h->group_info[(size_t)g * 4], b = h->group_info[(size_t)g * 4 + 1];
    if (a < 0 || b < 0 || a >= nb || b >= nb) continue;
    nz[(size_t)std::max(a, b)][(size_t)std::min(a, b)] = 1;
  }
  for (int k = 0; k < nb; k++)                                   // fill-in: rows i > j > k of column k couple (i, j)
    for (int i = k + 1; i < nb; i++)
      if (nz[(size_t)i][(size_t)k])
        for (int j = k + 1; j < i; j++)
          if (nz[(size_t)j][(size_t)k]) nz[(size_t)i][(size_t)j] = 1;
  h->blk_off.assign((size_t)nb * nb, -1);
  h->blk_rows.assign((size_t)nb, std::vector<int>());
  long off = 0;
  const long bsz = (long)h->nper * h->nper;
  for (int j = 0; j < nb; j++)
    for (int i = j; i < nb; i++)
      if (nz[(size_t)i][(size_t)j]) {
        h->blk_off[(size_t)i * nb + j] = off;
        off += bsz;
        if (i > j) h->blk_rows[(size_t)j].push_back(i);
      }
  return (int)(off / bsz);
}

static int factor_nonrigid_blocked(er_fopt_t h, size_t nv, size_t nd) {
  if (rocblas_load(h->roc, h->stream)) return 1;
  const int nb = h->num;
  const int B = h->nper;
  const long bsz = (long)B * B;
  const int n_blocks = block_symbolic(h);
  const size_t need = (size_t)n_blocks * (size_t)bsz;
  if (need > h->big_cap) {
    if (h->d_big) (void)hipFree(h->d_big);
    h->d_big = nullptr;
    h->big_cap = 0;
    hipError_t e = hipMalloc((void**)&h->d_big, need * sizeof(double));
    if (e != hipSuccess)
      return er::fail("er_fopt_factor_nonrigid: %d fragment blocks of %d x %d float64 (%.1f GB) do not fit: %s", n_blocks, B, B, (double)need * 8 / 1e9,
                      hipGetErrorString(e));
    h->big_cap = need;
  }
  if (h->d_blk_off) (void)hipFree(h->d_blk_off);
  h->d_blk_off = nullptr;
  ER_HIP_TRY(hipMalloc((void**)&h->d_blk_off, (size_t)nb * nb * sizeof(long)));
  ER_HIP_TRY(hipMemcpyAsync(h->d_blk_off, h->blk_off.data(), (size_t)nb * nb * sizeof(long), hipMemcpyHostToDevice, h->stream));
  ER_HIP_TRY(hipMemsetAsync(h->d_big, 0, need * sizeof(double), h->stream));
  const MatView V{h->d_big, 0, h->d_blk_off, nb, (long)B};
  hipLaunchKernelGGL(k_fopt_scatter_blocks, dim3((unsigned)((nd + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, h->d_diag, (long)(nd / 576),
                     (const int*)nullptr, 1, (int)nv, h->res, (long)h->nper, V);
  if (h->n_groups > 0)
    hipLaunchKernelGGL(k_fopt_scatter_blocks, dim3((unsigned)(((size_t)h->n_groups * 576 + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, h->d_off,
                       (long)h->n_groups, h->d_ginfo, 0, (int)nv, h->res, (long)h->nper, V);
  for (int l = 0; l < h->num; l++)                                            // baseAA, OptApp.cpp:765-810
    hipLaunchKernelGGL(k_fopt_add_laplacian_v, dim3((unsigned)((nv * 6 + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, V, (long)l * h->nper, h->res, 1.0);
  hipLaunchKernelGGL(k_fopt_add_diag_v, dim3(1), dim3(64), 0, h->stream, V, 0L, 3, 1.0);
  if (h->shift_index >= 0 && h->shift_index < (long)nb * B)
    hipLaunchKernelGGL(k_fopt_add_diag_v, dim3(1), dim3(64), 0, h->stream, V, h->shift_index, 1, h->shift_value);
  ER_HIP_TRY(hipGetLastError());
  // ---- numeric factorisation, right-looking over the blocks ----
  if (!h->d_info) ER_HIP_TRY(hipMalloc((void**)&h->d_info, sizeof(int)));
  if (h->d_rhs) {
    (void)hipFree(h->d_rhs);
    h->d_rhs = nullptr;
  }
  ER_HIP_TRY(hipMalloc((void**)&h->d_rhs, (size_t)nb * B * sizeof(double)));
  const double one = 1.0, minus = -1.0;
  auto blk = [&](int i, int j) { return h->d_big + h->blk_off[(size_t)i * nb + j]; };
  for (int k = 0; k < nb; k++) {
    if (potrf_lower(h, blk(k, k), B, B)) return 1;
    int info = 0;
    ER_HIP_TRY(hipMemcpyAsync(&info, h->d_info, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    ER_HIP_TRY(hipStreamSynchronize(h->stream));
    if (info != 0) {
      er::fail("the assembled system is not positive definite (Cholesky pivot %ld = fragment block %d, local pivot %d)", (long)k * B + info, k, info);
      return kNotPositiveDefinite;
    }
    const std::vector<int>& rows = h->blk_rows[(size_t)k];
    for (int i : rows)                                            // L_ik = A_ik L_kk^-T
      if (h->roc.dtrsm(h->roc.handle, kSideRight, kFillLower, kOpT, kDiagNonUnit, B, B, &one, blk(k, k), B, blk(i, k), B) != 0)
        return er::fail("rocblas_dtrsm failed (block %d,%d)", i, k);
    for (size_t a = 0; a < rows.size(); a++)                      // trailing update: A_ij -= L_ik L_jk^T
      for (size_t b = 0; b <= a; b++) {
        const int i = rows[a], j = rows[b];
        int rc;
        if (i == j)
          rc = h->roc.dsyrk(h->roc.handle, kFillLower, kOpN, B, B, &minus, blk(i, k), B, &one, blk(i, i), B);
        else
          rc = h->roc.dgemm(h->roc.handle, kOpN, kOpT, B, B, B, &minus, blk(i, k), B, blk(j, k), B, &one, blk(i, j), B);
        if (rc != 0) return er::fail("rocblas trailing update failed (block %d,%d)", i, j);
      }
  }
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  h->d_sys = h->d_big;
  h->sys_n = (long)nb * B;
  h->blocked = true;
  h->factored = true;
  return 0;
}

static int solve_blocked(er_fopt_t h) {                              // d_rhs <- (L L^T)^-1 d_rhs
  const int nb = h->num, B = h->nper;
  const double one = 1.0, minus = -1.0;
  auto blk = [&](int i, int j) { return h->d_big + h->blk_off[(size_t)i * nb + j]; };
  for (int k = 0; k < nb; k++) {                                     // forward: L y = b
    if (h->roc.dtrsv(h->roc.handle, kFillLower, kOpN, kDiagNonUnit, B, blk(k, k), B, h->d_rhs + (size_t)k * B, 1) != 0) return er::fail("rocblas_dtrsv failed");
    for (int i : h->blk_rows[(size_t)k])
      if (h->roc.dgemv(h->roc.handle, kOpN, B, B, &minus, blk(i, k), B, h->d_rhs + (size_t)k * B, 1, &one, h->d_rhs + (size_t)i * B, 1) != 0)
        return er::fail("rocblas_dgemv failed");
  }
  for (int k = nb - 1; k >= 0; k--) {                                // backward: L^T x = y
    for (int i : h->blk_rows[(size_t)k])
      if (h->roc.dgemv(h->roc.handle, kOpT, B, B, &minus, blk(i, k), B, h->d_rhs + (size_t)i * B, 1, &one, h->d_rhs + (size_t)k * B, 1) != 0)
        return er::fail("rocblas_dgemv failed");
    if (h->roc.dtrsv(h->roc.handle, kFillLower, kOpT, kDiagNonUnit, B, blk(k, k), B, h->d_rhs + (size_t)k * B, 1) != 0) return er::fail("rocblas_dtrsv failed");
  }
  return 0;
}

static int factor_nonrigid_once(er_fopt_t h, double weight);

int er_fopt_factor_nonrigid(er_fopt_t h, double weight) {
  return factor_nonrigid_once(h, weight) ? 1 : 0;
}

static int factor_nonrigid_once(er_fopt_t h, double weight) {
  if (!h) return er::fail("er_fopt_factor_nonrigid: NULL handle");
  ER_HIP_TRY(hipSetDevice(h->device));
  h->factored = false;
  h->blocked = false;
  const size_t nv = (size_t)h->nper / 3, M = (size_t)h->num * h->nper;
  const size_t nd = (size_t)h->num * nv * 576, no = (size_t)std::max(h->n_groups, 1) * 576;
  if (nd > h->diag_cap) {
    if (h->d_diag) (void)hipFree(h->d_diag);
    h->d_diag = nullptr;
    h->diag_cap = 0;
    ER_HIP_TRY(hipMalloc((void**)&h->d_diag, nd * sizeof(double)));
    h->diag_cap = nd;
  }
  if (no > h->off_cap) {
    if (h->d_off) (void)hipFree(h->d_off);
    h->d_off = nullptr;
    h->off_cap = 0;
    ER_HIP_TRY(hipMalloc((void**)&h->d_off, no * sizeof(double)));
    h->off_cap = no;
  }
  if (!h->d_ginfo && h->n_groups > 0) {
    ER_HIP_TRY(hipMalloc((void**)&h->d_ginfo, (size_t)h->n_groups * 4 * sizeof(int)));
    ER_HIP_TRY(hipMemcpyAsync(h->d_ginfo, h->group_info.data(), (size_t)h->n_groups * 4 * sizeof(int), hipMemcpyHostToDevice, h->stream));
  }
  ER_HIP_TRY(hipMemsetAsync(h->d_diag, 0, nd * sizeof(double), h->stream));
  ER_HIP_TRY(hipMemsetAsync(h->d_off, 0, no * sizeof(double), h->stream));
  ER_HIP_TRY(hipMemcpyAsync(h->d_rot, &weight, sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (h->n_chunks > 0)
    hipLaunchKernelGGL(k_fopt_gram<2>, dim3((h->n_chunks * 64 + kBlock - 1) / kBlock), dim3(kBlock), 0, h->stream, h->d_chunks, h->n_chunks,
                       h->d_frags, h->d_first, h->d_second, h->d_rot, h->num, h->res, 0, h->d_diag, h->d_off, (double*)nullptr);
  ER_HIP_TRY(hipGetLastError());
  // Dense while the square matrix is small (one potrf); block-sparse over fragments beyond ER_FOPT_DENSE_MAX unknowns
  // (default 30 000 = 7.2 GB dense; ER_FOPT_DENSE_MAX=0 forces the block path, used by the tests to compare the two).
  const char* env = getenv("ER_FOPT_DENSE_MAX");
  const size_t dense_max = env ? (size_t)atol(env) : 30000;
  if (M > dense_max) return factor_nonrigid_blocked(h, nv, nd);
  if (M * M > h->big_cap) {
    if (h->d_big) (void)hipFree(h->d_big);
    h->d_big = nullptr;
    h->big_cap = 0;
    hipError_t e = hipMalloc((void**)&h->d_big, M * M * sizeof(double));
    if (e != hipSuccess) return er::fail("er_fopt_factor_nonrigid: %zu x %zu float64 system (%.1f GB) does not fit: %s", M, M, (double)(M * M * 8) / 1e9, hipGetErrorString(e));
    h->big_cap = M * M;
  }
  ER_HIP_TRY(hipMemsetAsync(h->d_big, 0, M * M * sizeof(double), h->stream));
  const MatView V{h->d_big, (long)M, nullptr, 0, 0};
  hipLaunchKernelGGL(k_fopt_scatter_blocks, dim3((unsigned)((nd + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, h->d_diag, (long)(nd / 576),
                     (const int*)nullptr, 1, (int)nv, h->res, (long)h->nper, V);
  if (h->n_groups > 0)
    hipLaunchKernelGGL(k_fopt_scatter_blocks, dim3((unsigned)(((size_t)h->n_groups * 576 + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, h->d_off,
                       (long)h->n_groups, h->d_ginfo, 0, (int)nv, h->res, (long)h->nper, V);
  for (int l = 0; l < h->num; l++)                                            // baseAA, OptApp.cpp:765-810
    hipLaunchKernelGGL(k_fopt_add_laplacian, dim3((unsigned)((nv * 6 + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, h->d_big, (long)M,
                       (long)l * h->nper, h->res, 1.0);
  hipLaunchKernelGGL(k_fopt_add_diag, dim3(1), dim3(64), 0, h->stream, h->d_big, (long)M, 0L, 3, 1.0);
  ER_HIP_TRY(hipGetLastError());
  return factor_common(h, h->d_big, (long)M);
}

int er_fopt_debug_shift_diagonal(er_fopt_t h, long index, double value) {
  if (!h) return er::fail("er_fopt_debug_shift_diagonal: NULL handle");
  h->shift_index = index;
  h->shift_value = value;
  return 0;
}

int er_fopt_solve(er_fopt_t h, const double* rhs_host, int add_data_jb, double* x_host) {
  if (!h || !rhs_host || !x_host) return er::fail("er_fopt_solve: bad arguments");
  if (!h->factored) return er::fail("er_fopt_solve: no factored system (call er_fopt_factor_slac / er_fopt_factor_nonrigid first)");
  ER_HIP_TRY(hipSetDevice(h->device));
  const long n = h->sys_n;
  ER_HIP_TRY(hipMemcpyAsync(h->d_rhs, rhs_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (add_data_jb) {
    if (h->d_sys != h->d_JJ) return er::fail("er_fopt_solve: add_data_jb is only meaningful after er_fopt_factor_slac");
    hipLaunchKernelGGL(k_fopt_axpy, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, h->d_rhs, h->d_Jb, n);
    ER_HIP_TRY(hipGetLastError());
  }
  if (h->blocked) {
    if (solve_blocked(h)) return 1;
  } else if (h->roc.dtrsv(h->roc.handle, kFillLower, kOpN, kDiagNonUnit, (int)n, h->d_sys, (int)n, h->d_rhs, 1) != 0 ||       // L y = b
             h->roc.dtrsv(h->roc.handle, kFillLower, kOpT, kDiagNonUnit, (int)n, h->d_sys, (int)n, h->d_rhs, 1) != 0) {     // L^T x = y
    return er::fail("rocblas_dtrsv failed");
  }
  ER_HIP_TRY(hipMemcpyAsync(x_host, h->d_rhs, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

}  // extern "C"
