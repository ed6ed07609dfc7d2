// er_icp.hip -- path B of liber_hip.so: pairwise ICP refinement + correspondence building on MI355X
// (gfx950).  Replaces the numeric core of CCorresApp::Registration / FindCorrespondence
// (BuildCorrespondence/CorresApp.cpp:112-319) including the PCL pieces it calls
// (pcl::IterativeClosestPoint + TransformationEstimationPointToPlaneLLS + KdTreeFLANN, :295-312).
// PCL semantics follow SURVEY.md Appendix B (PCL 1.7; the library itself is not vendored -- see
// oracle/icp_oracle.cpp's header for what is assumed and DESIGN.md "parity unpinned").
//
// Data layout in HBM (one er_cloud_s per fragment, uploaded once, reused by every pair):
//   xyz, nrm     float[3n] in file order (the cloud as SOURCE; NaN-normal points already dropped)
//   sorted       float4[n] = {x, y, z, bit_cast(original index)} ordered by grid cell
//   cell_start   int[cells+1]   uniform grid, cell edge >= the largest search radius, so an exact
//                               nearest neighbour inside the radius lies in the 3x3x3 neighbourhood;
//                               cell id = (z*ny + y)*nx + x, so each (z,y) row is ONE contiguous range
// plus per-pair workspaces (stream, X, match, nd, pair list, sums, pinned result block) borrowed from a per-device
// pool, so clouds are immutable and any number of pairs can be in flight.
// Queries run in the SOURCE cloud's own cell-sorted order (thread t takes sorted[t]), so the lanes of a wave
// walk the same few target cells together (coalesced / broadcast candidate loads); results are written back
// by original index.  The NN search is block-cooperative (see nn_block); candidates stream from L2 as 16-byte loads:
//   k_count_inliers   transform (float64 -> float32) + NN + count           (Registration pre-check)
//   k_icp_iter        one ICP iteration: [apply last increment] + NN + point-to-plane rows -> 27+2 float64 sums
//                     (wave shuffle -> LDS -> per-workgroup partial -> fixed-order final sum by the last workgroup)
//   k_find_corr       transform points+normals + NN + distance/normal tests -> match[orig index];
//                     k_count_blocks (+ information-matrix sums) + k_scan_blocks + k_compact = stable compaction in file order
// Reductions and scans, not contractions: no MFMA.
#include "er_common.h"

#include "../../include/er_hip.h"

#include <hipcub/hipcub.hpp>   // device radix sort / prefix sum of the grid build (library primitives; everything else is hand-written)

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace {

constexpr int kBlock = 256;
constexpr int kAcc = 32;   // 21 ATA + 6 ATb + sum d^2 + count (+ padding)

struct Grid {
  const float4* pts;
  const int* cell_start;
  float org[3];
  float cell;
  int dim[3];
};

struct Mat12d { double m[12]; };
struct Mat12f { float m[12]; };

// Exact 1-NN of q among target points inside the 27 neighbouring cells: float32 squared distance
// ((dx*dx) + dy*dy) + dz*dz (FLANN L2_Simple), ties -> lower original index.  limit2 = squared search radius:
// callers discard anything farther, so rows of cells lying entirely beyond the radius are skipped (margin 1e-4
// relative for the float32 cell assignment).
//
// Block-cooperative, two phases (one query per thread, kBlock queries per workgroup):
//   phase 0  every thread scans the HOME row of its query (the three cells x-1..x+1 of its own (y,z) row, one
//            contiguous range) and learns a first best distance; the eight neighbouring rows that can still hold
//            a closer point (distance from q to the row's cell slab <= best so far) are appended to an LDS task
//            list as (query, row) pairs;
//   phase 1  the threads of the workgroup share the task list -- one row scan per thread and trip -- and fold the
//            results into the query's packed (distance bits, index) key with a 64-bit LDS atomicMin, which IS the
//            lexicographic (distance, index) minimum.
// Why: only ~1.7 of the 8 neighbour rows survive the test for an average query, but in a SIMT loop a wave runs
// every row that ANY of its 64 lanes needs -- practically all of them.  Compacting the surviving (query, row)
// pairs across the workgroup removes that waste (NN pass over 253 k queries: 34 -> 21 us).  Candidates are scanned
// kUnroll at a time (the 16-byte loads are issued together; the index is clamped to the row's last candidate,
// whose repeat cannot change the result).
#ifndef ER_ICP_UNROLL
#define ER_ICP_UNROLL 4
#endif
constexpr int kUnroll = ER_ICP_UNROLL;
constexpr unsigned long long kNoHit = ((unsigned long long)0x7f7fffffu << 32) | 0xffffffffull;   // (FLT_MAX, -1)

struct NnShared {
  unsigned long long best[kBlock];
  float q[3][kBlock];
  int x01[2][kBlock];
  int task_row[kBlock * 8];
  unsigned char task_q[kBlock * 8];
  int ntask;
};

__device__ __forceinline__ unsigned long long scan_row(const Grid& g, int row, int x0, int x1, float qx, float qy, float qz,
                                                       unsigned long long key) {
  const int s0 = g.cell_start[row + x0], s1 = g.cell_start[row + x1 + 1];
  for (int s = s0; s < s1; s += kUnroll) {
    float4 p[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) p[u] = g.pts[min(s + u, s1 - 1)];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
      const float dx = qx - p[u].x, dy = qy - p[u].y, dz = qz - p[u].z;
      const float d = ((dx * dx) + dy * dy) + dz * dz;
      // d >= 0, so its bit pattern orders like its value; NaN / inf patterns exceed FLT_MAX's and never win
      const unsigned long long k = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p[u].w);
      key = k < key ? k : key;
    }
  }
  return key;
}

// Every thread of the workgroup must call this (it synchronises); `active` = this thread carries a query.
// Returns the index (or -1) and the squared distance of the nearest target point.
__device__ __forceinline__ int nn_block(NnShared& sh, const Grid& g, bool active, float qx, float qy, float qz, float limit2,
                                        float& best_d) {
  const int tid = threadIdx.x;
  __syncthreads();                                            // the previous call's readers are done with `sh`
  if (tid == 0) sh.ntask = 0;
  __syncthreads();
  unsigned long long key = kNoHit;
  if (active) {
    const float ux = (qx - g.org[0]) / g.cell, uy = (qy - g.org[1]) / g.cell, uz = (qz - g.org[2]) / g.cell;
    const float cx = floorf(ux), cy = floorf(uy), cz = floorf(uz);
    const bool inside = cx >= -1.f && cx <= (float)g.dim[0] && cy >= -1.f && cy <= (float)g.dim[1] && cz >= -1.f && cz <= (float)g.dim[2];
    if (inside) {
      const int ix = (int)cx, iy = (int)cy, iz = (int)cz;
      const int x0 = max(ix - 1, 0), x1 = min(ix + 1, g.dim[0] - 1);
      if (x0 <= x1) {
        // distance from q to the lower / upper face of its own cell along y and z (metres)
        const float ylo = (uy - cy) * g.cell, yhi = g.cell - ylo, zlo = (uz - cz) * g.cell, zhi = g.cell - zlo;
        float bound = limit2 * 1.0001f + 1e-12f;
        if (iy >= 0 && iy < g.dim[1] && iz >= 0 && iz < g.dim[2]) {
          key = scan_row(g, (iz * g.dim[1] + iy) * g.dim[0], x0, x1, qx, qy, qz, key);
          bound = fminf(bound, __uint_as_float((unsigned)(key >> 32)) * 1.0001f + 1e-12f);   // neighbours must beat the home row
        }
        sh.q[0][tid] = qx;
        sh.q[1][tid] = qy;
        sh.q[2][tid] = qz;
        sh.x01[0][tid] = x0;
        sh.x01[1][tid] = x1;
#pragma unroll
        for (int pass = 0; pass < 9; pass++) {
          if (pass == 4) continue;
          const int dy = pass % 3 - 1, dz = pass / 3 - 1;
          const float ey = dy < 0 ? ylo : (dy > 0 ? yhi : 0.f), ez = dz < 0 ? zlo : (dz > 0 ? zhi : 0.f);
          const int y = iy + dy, z = iz + dz;
          if (y >= 0 && y < g.dim[1] && z >= 0 && z < g.dim[2] && ey * ey + ez * ez <= bound) {
            const int t = atomicAdd(&sh.ntask, 1);
            sh.task_row[t] = (z * g.dim[1] + y) * g.dim[0];
            sh.task_q[t] = (unsigned char)tid;
          }
        }
      }
    }
  }
  sh.best[tid] = key;
  __syncthreads();
  const int nt = sh.ntask;
  for (int t = tid; t < nt; t += kBlock) {
    const int q = sh.task_q[t];
    const unsigned long long k = scan_row(g, sh.task_row[t], sh.x01[0][q], sh.x01[1][q], sh.q[0][q], sh.q[1][q], sh.q[2][q], kNoHit);
    if (k != kNoHit) atomicMin(&sh.best[q], k);
  }
  __syncthreads();
  key = sh.best[tid];
  best_d = __uint_as_float((unsigned)(key >> 32));
  return (int)(unsigned)(key & 0xffffffffull);              // 0xffffffff -> -1
}

// Block reduction of NV float64 values per thread: wave shuffle (64 lanes) -> LDS -> lane 0 atomics.
template <int NV>
__device__ __forceinline__ void block_reduce_atomic(double (&v)[NV], double* __restrict__ out) {
  __shared__ double part[kBlock / 64][NV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; i++) {
    double s = v[i];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) part[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) s += part[w][threadIdx.x];
    if (s != 0.0) atomicAdd(&out[threadIdx.x], s);
  }
}

// pcl::transformPointCloudWithNormals with a Matrix4d: float64 evaluation, float32 storage.
__device__ __forceinline__ void xform_d(const Mat12d& T, float x, float y, float z, float& ox, float& oy, float& oz) {
  const double dx = x, dy = y, dz = z;
  ox = (float)(((T.m[0] * dx + T.m[1] * dy) + T.m[2] * dz) + T.m[3]);
  oy = (float)(((T.m[4] * dx + T.m[5] * dy) + T.m[6] * dz) + T.m[7]);
  oz = (float)(((T.m[8] * dx + T.m[9] * dy) + T.m[10] * dz) + T.m[11]);
}

// Registration pre-check, CorresApp.cpp:257-264.  A fixed grid strides over the points and issues ONE atomic per
// workgroup (one per wave serialised thousands of atomics on one word).
__global__ __launch_bounds__(kBlock) void k_count_inliers(const float4* __restrict__ src_sorted, int n, Mat12d T, Grid g, float radius,
                                                          double maxd2, int* __restrict__ count) {
  __shared__ NnShared sh;
  int local = 0;
  for (int base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {
    const int k = base + (int)threadIdx.x;
    float qx = 0.f, qy = 0.f, qz = 0.f, d;
    if (k < n) {
      const float4 s = src_sorted[k];
      xform_d(T, s.x, s.y, s.z, qx, qy, qz);
    }
    const int i = nn_block(sh, g, k < n, qx, qy, qz, radius * radius, d);
    if (k < n && i >= 0 && (double)d <= (double)radius * (double)radius && (double)d < maxd2) local++;
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
  __shared__ int part[kBlock / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < kBlock / 64; w++) s += part[w];
    if (s) atomicAdd(count, s);
  }
}

// RansacCurvature::getFitness (GlobalRegistration/RansacCurvature.h:661-704) for MANY pose hypotheses of one
// (source, target) pair: blockIdx.y = hypothesis, blockIdx.x strides over the source points.  Float32 transform
// ([PCL] transformPointCloud with a Matrix4f), exact NN, inlier iff d < threshold^2 (float compare, :670,:687);
// per hypothesis the inlier count (exact) and the float64 sum of the inlier distances.
__global__ __launch_bounds__(kBlock) void k_ransac_fitness(const float4* __restrict__ src_sorted, int n, const float* __restrict__ hyp,
                                                           int hyp0, Grid g, float radius, float max_range,
                                                           int* __restrict__ count, double* __restrict__ sum) {
  __shared__ NnShared sh;
  const int h = hyp0 + blockIdx.y;
  float M[12];
#pragma unroll
  for (int q = 0; q < 12; q++) M[q] = hyp[(size_t)h * 16 + q];
  int local = 0;
  double dsum = 0.0;
  for (int base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {
    const int k = base + (int)threadIdx.x;
    float qx = 0.f, qy = 0.f, qz = 0.f, d;
    if (k < n) {
      const float4 s = src_sorted[k];
      qx = ((M[0] * s.x + M[1] * s.y) + M[2] * s.z) + M[3];
      qy = ((M[4] * s.x + M[5] * s.y) + M[6] * s.z) + M[7];
      qz = ((M[8] * s.x + M[9] * s.y) + M[10] * s.z) + M[11];
    }
    const int i = nn_block(sh, g, k < n, qx, qy, qz, radius * radius, d);
    if (k < n && i >= 0 && d < max_range) {
      local++;
      dsum += (double)d;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    local += __shfl_down(local, off);
    dsum += __shfl_down(dsum, off);
  }
  __shared__ int pc[kBlock / 64];
  __shared__ double ps[kBlock / 64];
  if ((threadIdx.x & 63) == 0) {
    pc[threadIdx.x >> 6] = local;
    ps[threadIdx.x >> 6] = dsum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
    double s = 0.0;
    for (int w = 0; w < kBlock / 64; w++) {
      c += pc[w];
      s += ps[w];
    }
    if (c) {
      atomicAdd(&count[h], c);
      atomicAdd(&sum[h], s);
    }
  }
}

// The accepted hypothesis once more, this time keeping the lists (RansacCurvature.h:661-704: `inliers`, `inliers_target`):
// match[original source index] = NN index if d < threshold^2, else -1; acc[20] += d over the inliers.
__global__ __launch_bounds__(kBlock) void k_ransac_match(const float4* __restrict__ src_sorted, int n, Mat12f M, Grid g, float radius,
                                                         float max_range, int* __restrict__ match, double* __restrict__ acc) {
  __shared__ NnShared sh;
  const int q = blockIdx.x * kBlock + threadIdx.x;
  float qx = 0.f, qy = 0.f, qz = 0.f, d;
  int k = 0;
  if (q < n) {
    const float4 s = src_sorted[q];
    k = __float_as_int(s.w);
    qx = ((M.m[0] * s.x + M.m[1] * s.y) + M.m[2] * s.z) + M.m[3];
    qy = ((M.m[4] * s.x + M.m[5] * s.y) + M.m[6] * s.z) + M.m[7];
    qz = ((M.m[8] * s.x + M.m[9] * s.y) + M.m[10] * s.z) + M.m[11];
  }
  const int i = nn_block(sh, g, q < n, qx, qy, qz, radius * radius, d);
  const bool hit = q < n && i >= 0 && d < max_range;
  if (q < n) match[k] = hit ? i : -1;
  double v[1] = {hit ? (double)d : 0.0};
  __syncthreads();
  block_reduce_atomic<1>(v, acc + 20);
}

// getInformation's target half (RansacCurvature.h:723-731): the ten distinct terms of sum A^T A (see k_count_blocks) over the
// matched TARGET points.
__global__ __launch_bounds__(kBlock) void k_info_matched(const int* __restrict__ match, const float* __restrict__ tgt_xyz, int n,
                                                         double* __restrict__ info) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  double v[10];
#pragma unroll
  for (int i = 0; i < 10; i++) v[i] = 0.0;
  const int m = k < n ? match[k] : -1;
  if (m >= 0) {
    const float tx = tgt_xyz[3 * m], ty = tgt_xyz[3 * m + 1], tz = tgt_xyz[3 * m + 2];
    const double ax = (double)(2 * tx), ay = (double)(2 * ty), az = (double)(2 * tz);
    v[0] = ax; v[1] = ay; v[2] = az;
    v[3] = az * az + ay * ay;
    v[4] = az * az + ax * ax;
    v[5] = ay * ay + ax * ax;
    v[6] = ay * (-ax);
    v[7] = (-az) * ax;
    v[8] = az * (-ay);
    v[9] = 1.0;
  }
  block_reduce_atomic<10>(v, info);
}

// guess * source in float32 (IterativeClosestPoint::transformCloud), or a plain copy for an identity guess.
__global__ void k_init_x(const float4* __restrict__ src_sorted, float* __restrict__ X, int n, Mat12f M, int apply) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;       // k = position in the source's cell-sorted order
  if (k >= n) return;
  const float4 s = src_sorted[k];
  const float x = s.x, y = s.y, z = s.z;
  if (apply) {
    X[3 * k] = ((M.m[0] * x + M.m[1] * y) + M.m[2] * z) + M.m[3];
    X[3 * k + 1] = ((M.m[4] * x + M.m[5] * y) + M.m[6] * z) + M.m[7];
    X[3 * k + 2] = ((M.m[8] * x + M.m[9] * y) + M.m[10] * z) + M.m[11];
  } else {
    X[3 * k] = x;
    X[3 * k + 1] = y;
    X[3 * k + 2] = z;
  }
}

// ---- the ICP loop's state lives on the device ------------------------------------------------------------------------
// pcl::IterativeClosestPoint::align's loop variables (final_transformation_, the last increment, the previous MSE, the
// iteration counter and the convergence flags).  k_icp_final updates them from the iteration's sums -- 6x6 solve, increment,
// PCL's stop rule -- and k_icp_iter reads the increment (and the `done` flag) from here, so a whole chunk of iterations is
// enqueued without a host round trip; launches that find `done` set return at once.
struct IcpDev {
  float fin[16];           // final_transformation_
  float delta[16];         // last increment (identity before the first solve)
  float prev_delta[16];
  double prev_mse;
  int iter, done, conv, apply;   // apply: the next k_icp_iter multiplies X by delta first
};

struct IcpParams {
  double eps;
  int max_iter, stop_rule;
};

// Dense 6x6 solve by Gaussian elimination with partial pivoting (PCL: ATA.inverse() * ATb).
__device__ bool dev_solve6x6(double A[6][6], double b[6], double x[6]) {
  for (int c = 0; c < 6; c++) {
    int p = c;
    for (int r = c + 1; r < 6; r++)
      if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
    if (A[p][c] == 0.0 || !isfinite(A[p][c])) return false;
    if (p != c) {
      for (int k = 0; k < 6; k++) { const double t = A[p][k]; A[p][k] = A[c][k]; A[c][k] = t; }
      const double t = b[p]; b[p] = b[c]; b[c] = t;
    }
    for (int r = c + 1; r < 6; r++) {
      const double f = A[r][c] / A[c][c];
      for (int k = c; k < 6; k++) A[r][k] -= f * A[c][k];
      b[r] -= f * b[c];
    }
  }
  for (int r = 5; r >= 0; r--) {
    double s = b[r];
    for (int k = r + 1; k < 6; k++) s -= A[r][k] * x[k];
    x[r] = s / A[r][r];
  }
  return true;
}

// The same system by an LDL^T factorisation with compile-time indices (everything stays in registers; the pivoting
// elimination above indexes its rows dynamically, which puts the matrix into scratch memory: ~10 us for one thread).  A^T A is
// symmetric positive definite whenever the correspondences constrain all six degrees of freedom; returns false (caller
// falls back to the pivoting elimination) as soon as a pivot is not safely positive.
__device__ bool dev_solve6x6_spd(const double* acc, double x[6]) {
  double a[6][6];
  {
    int t = 0;
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = r; c < 6; c++) a[r][c] = a[c][r] = acc[t++];
  }
  double d[6], y[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {                               // a[i][j] (i > j) becomes L_ij, d[j] the pivot
    double dj = a[j][j];
#pragma unroll
    for (int k = 0; k < j; k++) dj -= a[j][k] * a[j][k] * d[k];
    if (!(dj > 1e-300) || !isfinite(dj)) return false;
    d[j] = dj;
    const double inv = 1.0 / dj;
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      double s = a[i][j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= a[i][k] * a[j][k] * d[k];
      a[i][j] = s * inv;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {                               // L y = b
    double s = acc[21 + i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= a[i][k] * y[k];
    y[i] = s;
  }
#pragma unroll
  for (int i = 5; i >= 0; i--) {                              // L^T x = D^-1 y
    double s = y[i] / d[i];
#pragma unroll
    for (int k = i + 1; k < 6; k++) s -= a[k][i] * x[k];
    x[i] = s;
  }
  bool ok = true;
#pragma unroll
  for (int i = 0; i < 6; i++) ok = ok && isfinite(x[i]);
  return ok;
}

// TransformationEstimationPointToPlaneLLS::constructTransformationMatrix: Rz(gamma) Ry(beta) Rx(alpha), float storage.
__device__ void dev_construct_increment(const double x[6], float M[16]) {
  double sa, ca, sb, cb, sg, cg;
  sincos(x[0], &sa, &ca);
  sincos(x[1], &sb, &cb);
  sincos(x[2], &sg, &cg);
  for (int i = 0; i < 16; i++) M[i] = 0.f;
  M[0] = (float)(cg * cb);
  M[1] = (float)(-sg * ca + cg * sb * sa);
  M[2] = (float)(sg * sa + cg * sb * ca);
  M[4] = (float)(sg * cb);
  M[5] = (float)(cg * ca + sg * sb * sa);
  M[6] = (float)(-cg * sa + sg * sb * ca);
  M[8] = (float)(-sb);
  M[9] = (float)(cb * sa);
  M[10] = (float)(cb * ca);
  M[3] = (float)x[3];
  M[7] = (float)x[4];
  M[11] = (float)x[5];
  M[15] = 1.f;
}

// One step of IterativeClosestPoint's loop after the correspondences of the iteration have been summed (acc[0..28]):
// min_number_correspondences_, estimateRigidTransformation, final = increment * final, ++iterations, the stop rule.
__device__ void dev_icp_decide(const double* acc, IcpDev* st, IcpParams P) {
  const double cnt = acc[28];
  double A[6][6], b[6], x[6];
  bool stop = false;
  st->apply = 0;
  if (cnt < 3.0) {                                           // min_number_correspondences_
    st->conv = 0;
    stop = true;
  } else {
    if (!dev_solve6x6_spd(acc, x)) {                         // (not positive definite: the general elimination decides)
      int t = 0;
      for (int r = 0; r < 6; r++)
        for (int c2 = r; c2 < 6; c2++) A[r][c2] = A[c2][r] = acc[t++];
      for (int r = 0; r < 6; r++) b[r] = acc[21 + r];
      if (!dev_solve6x6(A, b, x)) {
        st->conv = 0;
        stop = true;
      }
    }
  }
  if (!stop) {
    float D[16], F[16];
    for (int i = 0; i < 16; i++) st->prev_delta[i] = st->delta[i];
    dev_construct_increment(x, D);
    for (int r = 0; r < 4; r++)                              // final = increment * final
      for (int c = 0; c < 4; c++)
        F[r * 4 + c] = ((D[r * 4] * st->fin[c] + D[r * 4 + 1] * st->fin[4 + c]) + D[r * 4 + 2] * st->fin[8 + c]) + D[r * 4 + 3] * st->fin[12 + c];
    for (int i = 0; i < 16; i++) {
      st->fin[i] = F[i];
      st->delta[i] = D[i];
    }
    st->iter += 1;
    st->apply = 1;
    if (st->iter >= P.max_iter) {
      st->conv = 1;
      stop = true;
    } else if (P.stop_rule == 0) {                           // PCL 1.7 DefaultConvergenceCriteria
      const double cos_angle = 0.5 * (double)(D[0] + D[5] + D[10] - 1.f);
      const double tr2 = (double)(D[3] * D[3] + D[7] * D[7] + D[11] * D[11]);
      const double cur = acc[27] / cnt;
      if (cos_angle >= 1.0 - P.eps && tr2 <= P.eps) {
        st->conv = 1;
        stop = true;
      } else if (fabs(cur - st->prev_mse) < 1e-12) {
        st->conv = 1;
        stop = true;
      }
      st->prev_mse = cur;
    } else {                                                 // PCL <= 1.6
      float sum = 0.f;
      for (int i = 0; i < 16; i++) sum += D[i] - st->prev_delta[i];
      if (fabs((double)sum) < P.eps) {
        st->conv = 1;
        stop = true;
      }
    }
  }
  if (stop) st->done = 1;
}

// One ICP iteration in one launch:
//   X <- delta * X (the previous iteration's increment, float32); correspondence estimation (exact NN, kept if
//   d^2 <= max_dist^2); the sums of TransformationEstimationPointToPlaneLLS over the kept correspondences:
//   acc[0..20] upper triangle of AtA row by row, [21..26] Atb, [27] sum of d^2, [28] count.
// Reduction: thread rows -> wave shuffle tree -> LDS -> ONE partial vector per workgroup in `partial`; k_icp_final adds
// the partial vectors in a fixed order.  No float64 atomics: the sums are bit-reproducible from run to run (they
// still differ from a sequential CPU sum in the last bits).
__global__ __launch_bounds__(kBlock) void k_icp_iter(float* __restrict__ X, int n, const IcpDev* __restrict__ st, Grid g, float radius,
                                                     double maxd2, const float* __restrict__ tgt_xyz, const float* __restrict__ tgt_nrm,
                                                     double* __restrict__ partial) {
  __shared__ NnShared sh;
  if (st->done) return;                                      // the loop has ended: this launch of the chunk is a no-op
  const int apply = st->apply;                               // wave-uniform (scalar loads)
  const float* __restrict__ dm = st->delta;
  const int k = blockIdx.x * kBlock + threadIdx.x;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  if (k < n) {
    sx = X[3 * k], sy = X[3 * k + 1], sz = X[3 * k + 2];
    if (apply) {
      const float x = sx, y = sy, z = sz;
      sx = ((dm[0] * x + dm[1] * y) + dm[2] * z) + dm[3];
      sy = ((dm[4] * x + dm[5] * y) + dm[6] * z) + dm[7];
      sz = ((dm[8] * x + dm[9] * y) + dm[10] * z) + dm[11];
      X[3 * k] = sx;
      X[3 * k + 1] = sy;
      X[3 * k + 2] = sz;
    }
  }
  float d;
  const int i = nn_block(sh, g, k < n, sx, sy, sz, radius * radius, d);
  double v[29];
#pragma unroll
  for (int t = 0; t < 29; t++) v[t] = 0.0;
  if (k < n && i >= 0 && (double)d <= (double)radius * (double)radius && !((double)d > maxd2)) {
    const float dx = tgt_xyz[3 * i], dy = tgt_xyz[3 * i + 1], dz = tgt_xyz[3 * i + 2];
    const float nx = tgt_nrm[3 * i], ny = tgt_nrm[3 * i + 1], nz = tgt_nrm[3 * i + 2];
    v[27] = (double)d;
    v[28] = 1.0;
    if (isfinite(sx) && isfinite(sy) && isfinite(sz) && isfinite(nx) && isfinite(ny) && isfinite(nz)) {
      const double a = (double)(nz * sy - ny * sz);      // float32 expressions widened to double (PCL)
      const double b = (double)(nx * sz - nz * sx);
      const double c = (double)(ny * sx - nx * sy);
      const double dnx = nx, dny = ny, dnz = nz;
      v[0] = a * a;  v[1] = a * b;  v[2] = a * c;  v[3] = a * dnx;  v[4] = a * dny;  v[5] = a * dnz;
      v[6] = b * b;  v[7] = b * c;  v[8] = b * dnx; v[9] = b * dny; v[10] = b * dnz;
      v[11] = c * c; v[12] = c * dnx; v[13] = c * dny; v[14] = c * dnz;
      v[15] = dnx * dnx; v[16] = dnx * dny; v[17] = dnx * dnz;
      v[18] = dny * dny; v[19] = dny * dnz;
      v[20] = dnz * dnz;
      const double e = (double)(nx * dx + ny * dy + nz * dz - nx * sx - ny * sy - nz * sz);
      v[21] = a * e; v[22] = b * e; v[23] = c * e; v[24] = dnx * e; v[25] = dny * e; v[26] = dnz * e;
    }
  }
  // workgroup partial
  __shared__ double part[kBlock / 64][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 29; t++) {
    double q = v[t];
    for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off);
    if (lane == 0) part[wave][t] = q;
  }
  __syncthreads();
  if (threadIdx.x < 29) {
    double q = 0.0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; w++) q += part[w][threadIdx.x];
    partial[(size_t)blockIdx.x * 32 + threadIdx.x] = q;
  }
}

// Second half of the reduction: ONE workgroup adds the per-workgroup partial vectors in a fixed order (8 strided
// slices per value, then the slices in order) -> acc[0..28].  A separate launch on purpose: finishing inside
// k_icp_iter ("last workgroup done" ticket + __threadfence) costs an L2 write-back per workgroup on this
// multi-XCD part (measured: 100-600 us per launch instead of ~25).
__global__ __launch_bounds__(kBlock) void k_icp_final(const double* __restrict__ partial, int nparts, double* __restrict__ acc,
                                                      IcpDev* __restrict__ st, IcpParams P) {
  if (st->done) return;
  const int val = threadIdx.x & 31, slice = threadIdx.x >> 5;   // 32 x 8
  double q = 0.0;
  if (val < 29) {
#pragma unroll 8
    for (int b = slice; b < nparts; b += 8) q += partial[(size_t)b * 32 + val];
  }
  __shared__ double fin[8][32];
  fin[slice][val] = q;
  __syncthreads();
  __shared__ double tot[32];
  if (threadIdx.x < 29) {
    double r = 0.0;
#pragma unroll
    for (int sl = 0; sl < 8; sl++) r += fin[sl][threadIdx.x];
    acc[threadIdx.x] = r;
    tot[threadIdx.x] = r;
  }
  __syncthreads();
  if (threadIdx.x == 0) dev_icp_decide(tot, st, P);          // solve, increment, stop rule: the loop never leaves the device
}

// getFitnessScore-style diagnostic: squared NN distance of final * source inside the search radius (-1 = none).
__global__ __launch_bounds__(kBlock) void k_fitness_nn(const float4* __restrict__ src_sorted, int n, Mat12f M, Grid g, float radius,
                                                       float* __restrict__ nd) {
  __shared__ NnShared sh;
  const int k = blockIdx.x * kBlock + threadIdx.x;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (k < n) {
    const float4 s = src_sorted[k];
    const float x = s.x, y = s.y, z = s.z;
    qx = ((M.m[0] * x + M.m[1] * y) + M.m[2] * z) + M.m[3];
    qy = ((M.m[4] * x + M.m[5] * y) + M.m[6] * z) + M.m[7];
    qz = ((M.m[8] * x + M.m[9] * y) + M.m[10] * z) + M.m[11];
  }
  float d;
  const int i = nn_block(sh, g, k < n, qx, qy, qz, radius * radius, d);
  if (k < n) nd[k] = (i >= 0 && (double)d <= (double)radius * (double)radius) ? d : -1.0f;
}

__global__ __launch_bounds__(kBlock) void k_fitness_sum(const float* __restrict__ nd, int n, double* __restrict__ acc) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  double v[2] = {0.0, 0.0};
  if (k < n && nd[k] >= 0.0f) {
    v[0] = (double)nd[k];
    v[1] = 1.0;
  }
  block_reduce_atomic<2>(v, acc);
}

// FindCorrespondence, CorresApp.cpp:144-161: match[original index] = NN index
// passing the distance and normal tests, else -1.
__global__ __launch_bounds__(kBlock) void k_find_corr(const float4* __restrict__ src_sorted, const float* __restrict__ nrm, int n,
                                                      Mat12d T, Grid g, const float* __restrict__ tgt_nrm, float radius,
                                                      double dist2, double normal_cos, int* __restrict__ match) {
  __shared__ NnShared sh;
  const int q = blockIdx.x * kBlock + threadIdx.x;             // q = position in the source's cell-sorted order
  float qx = 0.f, qy = 0.f, qz = 0.f, d;
  int k = 0;
  if (q < n) {
    const float4 s = src_sorted[q];
    k = __float_as_int(s.w);                                   // original (file-order) index of this source point
    xform_d(T, s.x, s.y, s.z, qx, qy, qz);
  }
  const int i = nn_block(sh, g, q < n, qx, qy, qz, radius * radius, d);
  if (q >= n) return;
  int m = -1;
  if (i >= 0 && (double)d <= (double)radius * (double)radius && (double)d < dist2) {         // :154
    const double nx = nrm[3 * k], ny = nrm[3 * k + 1], nz = nrm[3 * k + 2];
    const float tnx = (float)((T.m[0] * nx + T.m[1] * ny) + T.m[2] * nz);                     // n' = R n (double -> float)
    const float tny = (float)((T.m[4] * nx + T.m[5] * ny) + T.m[6] * nz);
    const float tnz = (float)((T.m[8] * nx + T.m[9] * ny) + T.m[10] * nz);
    // NormalDot, CorresApp.h:58-60: float32 products/sums, compared as double
    const float dot = (tgt_nrm[3 * i] * tnx + tgt_nrm[3 * i + 1] * tny) + tgt_nrm[3 * i + 2] * tnz;
    if ((double)dot > normal_cos) m = i;                                                       // :155
  }
  match[k] = m;
}

// Matches per block of 256 consecutive ORIGINAL indices (the order of the output list) and, optionally, the
// information matrix of CorresApp.cpp:186-208 over the matched, UNtransformed source points:
// info[0..2] = sum 2sx,2sy,2sz; [3..5] = sum (4sz^2+4sy^2),(4sz^2+4sx^2),(4sy^2+4sx^2);
// [6..8] = sum -4sysx, -4szsx, -4szsy; [9] = count   (the distinct terms of sum A^T A, A = [I | 2*skew-like(s)])
__global__ __launch_bounds__(kBlock) void k_count_blocks(const int* __restrict__ match, const float* __restrict__ xyz, int n,
                                                         int* __restrict__ block_count, double* __restrict__ info, int want_info) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const bool hit = k < n && match[k] >= 0;
  const unsigned long long b = __ballot(hit);
  __shared__ int wcnt[kBlock / 64];
  if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < kBlock / 64; w++) s += wcnt[w];
    block_count[blockIdx.x] = s;
  }
  if (want_info) {
    double v[10];
#pragma unroll
    for (int i = 0; i < 10; i++) v[i] = 0.0;
    if (hit) {                                                                                 // :192-204
      const float sx = xyz[3 * k], sy = xyz[3 * k + 1], sz = xyz[3 * k + 2];
      const double ax = (double)(2 * sx), ay = (double)(2 * sy), az = (double)(2 * sz);
      v[0] = ax; v[1] = ay; v[2] = az;
      v[3] = az * az + ay * ay;     // (0*0 + (-2sz)(-2sz)) + (2sy)(2sy)
      v[4] = az * az + ax * ax;     // ((2sz)(2sz) + 0*0) + (-2sx)(-2sx)
      v[5] = ay * ay + ax * ax;     // ((-2sy)(-2sy) + (2sx)(2sx)) + 0*0
      v[6] = ay * (-ax);            // (3,4): (2sy)(-2sx)
      v[7] = (-az) * ax;            // (3,5): (-2sz)(2sx)
      v[8] = az * (-ay);            // (4,5): (2sz)(-2sy)
      v[9] = 1.0;
    }
    __syncthreads();
    block_reduce_atomic<10>(v, info);
  }
}

// Exclusive scan of the per-block match counts (a few thousand blocks at most): one workgroup.
__global__ __launch_bounds__(1024) void k_scan_blocks(const int* __restrict__ block_count, int* __restrict__ block_offset, int nb,
                                                       int* __restrict__ total) {
  __shared__ int buf[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + (int)threadIdx.x;
    const int v = i < nb ? block_count[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int t = threadIdx.x >= (unsigned)off ? buf[threadIdx.x - off] : 0;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) block_offset[i] = carry + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += buf[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

// Stable compaction: pairs (match[k], k) in ascending k (CorresApp.cpp:157, file order of corres_*.txt).
__global__ __launch_bounds__(kBlock) void k_compact(const int* __restrict__ match, int n, const int* __restrict__ block_offset,
                                                    int* __restrict__ pairs, int capacity) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = k < n ? match[k] : -1;
  const unsigned long long b = __ballot(m >= 0);
  __shared__ int wcnt[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wcnt[wave] = __popcll(b);
  __syncthreads();
  if (m >= 0) {
    int o = block_offset[blockIdx.x] + __popcll(b & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; w++) o += wcnt[w];
    if (o < capacity) {
      pairs[2 * o] = m;
      pairs[2 * o + 1] = k;
    }
  }
}

// ---- uniform grid of a cloud, built on the device ---------------------------------------------------------------------
// (the reference builds a kd-tree per pair and per function, CorresApp.cpp:129,238; here once per fragment)
//   k_grid_bounds   min / max of the coordinates (float bits mapped to ordered ints, block reduce, 6 atomics per block) + a
//                   non-finite flag;
//   k_grid_cells    cell id of every point (the same float32 expression nn_block evaluates for a query) + histogram;
//   hipcub          stable radix sort of (cell id, original index) and the prefix sum of the histogram -> cell_start;
//   k_grid_gather   sorted[s] = {x, y, z, original index}.
// A stable sort keeps the points of a cell in file order, like a counting sort on the host would: the layout -- and with it the
// order of every float64 reduction that walks the cloud -- is reproducible from run to run.
__device__ __forceinline__ int ordered_int(float f) {
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ __forceinline__ float ordered_float(int i) {
  const int b = i >= 0 ? i : i ^ 0x7fffffff;
  float f;
  memcpy(&f, &b, sizeof f);
  return f;
}

__global__ __launch_bounds__(kBlock) void k_grid_bounds(const float* __restrict__ xyz, int n, int* __restrict__ out7) {
  int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
  int bad = 0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float v = xyz[3 * (size_t)i + a];
      bad |= !isfinite(v);
      const int o = ordered_int(v);
      lo[a] = min(lo[a], o);
      hi[a] = max(hi[a], o);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++)
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = min(lo[a], __shfl_down(lo[a], off));
      hi[a] = max(hi[a], __shfl_down(hi[a], off));
    }
  bad = __any(bad) ? 1 : 0;
  __shared__ int part[kBlock / 64][7];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      part[wave][a] = lo[a];
      part[wave][3 + a] = hi[a];
    }
    part[wave][6] = bad;
  }
  __syncthreads();
  if (threadIdx.x < 7) {                                      // 7 atomics per workgroup (one per wave serialised on 6 words: 245 us)
    int v = part[0][threadIdx.x];
    for (int w = 1; w < kBlock / 64; w++)
      v = threadIdx.x < 3 ? min(v, part[w][threadIdx.x]) : (threadIdx.x < 6 ? max(v, part[w][threadIdx.x]) : (v | part[w][threadIdx.x]));
    if (threadIdx.x < 3) atomicMin(&out7[threadIdx.x], v);
    else if (threadIdx.x < 6) atomicMax(&out7[threadIdx.x], v);
    else if (v) atomicOr(&out7[6], 1);
  }
}

struct GridDims {
  float org[3];
  float cell;
  int dim[3];
};

__global__ __launch_bounds__(kBlock) void k_grid_cells(const float* __restrict__ xyz, int n, GridDims G, unsigned* __restrict__ key,
                                                       unsigned* __restrict__ idx, int* __restrict__ count) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  int q[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    q[a] = (int)floorf((xyz[3 * (size_t)i + a] - G.org[a]) / G.cell);
    q[a] = min(max(q[a], 0), G.dim[a] - 1);
  }
  const int c = (q[2] * G.dim[1] + q[1]) * G.dim[0] + q[0];
  key[i] = (unsigned)c;
  idx[i] = (unsigned)i;
  atomicAdd(&count[c + 1], 1);                               // integer histogram: the result does not depend on the order
}

__global__ __launch_bounds__(kBlock) void k_grid_gather(const float* __restrict__ xyz, const unsigned* __restrict__ idx, int n,
                                                        float4* __restrict__ sorted) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= n) return;
  const unsigned i = idx[s];
  sorted[s] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __int_as_float((int)i));
}

// Grow-only scratch of the grid build, one per device, handed out under a mutex (er_cloud_create may be called from
// several host threads; builds on one device then take turns).
struct GridScratch {
  std::mutex mu;
  int device = -1;
  unsigned *key[2] = {nullptr, nullptr}, *idx[2] = {nullptr, nullptr};
  void* cub = nullptr;
  int* bounds = nullptr;
  size_t n_cap = 0, cub_cap = 0;
};
GridScratch& grid_scratch(int device) {
  static GridScratch* tab = new GridScratch[64];             // intentionally leaked (see WsPool)
  return tab[device & 63];
}

}  // namespace

struct er_cloud_s {
  int device = 0, n = 0;
  float *xyz = nullptr, *nrm = nullptr;
  float4* sorted = nullptr;
  int* cell_start = nullptr;
  Grid grid{};
  float radius_cap = 0.f;       // largest search radius the grid supports
};

namespace {
Grid grid_of(const er_cloud_s* c) { return c->grid; }

// ---- workspaces --------------------------------------------------------------------------------
// Everything a pair needs while it is being processed (clouds are immutable and shared): one HIP stream, the
// per-source-point scratch and a small pinned block the kernels' results are copied into.  Workspaces live in
// a per-device pool: a single-pair call borrows one, a *_batch call borrows several and software-pipelines
// its pairs over them so that one pair's host round trip (6x6 solve, convergence test) overlaps the kernels
// of the others.  The pool is never freed behind the HIP runtime's back (er_icp_release_workspaces does it).
struct HostBlock {
  double acc[kAcc];
  int count[4];
  IcpDev state;                 // the ICP loop's state: uploaded at the start of a job, read back after every chunk of iterations
};

struct IcpWs {
  int device = 0;
  size_t cap = 0;               // points
  hipStream_t stream = nullptr;
  hipEvent_t ev = nullptr;
  float *X = nullptr, *nd = nullptr;
  int *match = nullptr, *block_count = nullptr, *block_offset = nullptr, *pairs = nullptr, *icount = nullptr;
  double* acc = nullptr;
  IcpDev* dstate = nullptr;     // the ICP loop's state on the device
  double* partial = nullptr;    // one 32-double vector per workgroup of k_icp_iter
  HostBlock* host = nullptr;    // pinned
  int* stage = nullptr;         // pinned, cap * 2 ints: pair lists on their way to pageable caller memory (lazy)
  size_t stage_cap = 0;
};

void ws_free_buffers(IcpWs* w) {
  void* ptrs[] = {w->X, w->nd, w->match, w->block_count, w->block_offset, w->pairs, w->partial};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  w->partial = nullptr;
  w->X = w->nd = nullptr;
  w->match = w->block_count = w->block_offset = w->pairs = nullptr;
  w->cap = 0;
}

void ws_destroy(IcpWs* w) {
  if (!w) return;
  (void)hipSetDevice(w->device);
  if (w->stream) (void)hipStreamSynchronize(w->stream);
  ws_free_buffers(w);
  if (w->icount) (void)hipFree(w->icount);
  if (w->acc) (void)hipFree(w->acc);
  if (w->dstate) (void)hipFree(w->dstate);
  if (w->host) (void)hipHostFree(w->host);
  if (w->stage) (void)hipHostFree(w->stage);
  if (w->ev) (void)hipEventDestroy(w->ev);
  if (w->stream) (void)hipStreamDestroy(w->stream);
  delete w;
}

int ws_reserve(IcpWs* w, size_t n) {
  n = std::max<size_t>(n, 1);
  if (n <= w->cap) return 0;
  ER_HIP_TRY(hipStreamSynchronize(w->stream));
  ws_free_buffers(w);
  const size_t cap = n + n / 8;
  const size_t nb = (cap + kBlock - 1) / kBlock;
  ER_HIP_TRY(hipMalloc((void**)&w->X, cap * 3 * sizeof(float)));
  ER_HIP_TRY(hipMalloc((void**)&w->nd, cap * sizeof(float)));
  ER_HIP_TRY(hipMalloc((void**)&w->match, cap * sizeof(int)));
  ER_HIP_TRY(hipMalloc((void**)&w->pairs, cap * 2 * sizeof(int)));
  ER_HIP_TRY(hipMalloc((void**)&w->block_count, nb * sizeof(int)));
  ER_HIP_TRY(hipMalloc((void**)&w->block_offset, nb * sizeof(int)));
  ER_HIP_TRY(hipMalloc((void**)&w->partial, nb * 32 * sizeof(double)));
  w->cap = cap;
  return 0;
}

struct WsPool {
  std::mutex mu;
  std::vector<IcpWs*> idle;
};
WsPool& pool() {
  static WsPool* p = new WsPool();      // intentionally leaked: must outlive every static destructor
  return *p;
}

IcpWs* ws_acquire(int device, size_t n) {
  IcpWs* w = nullptr;
  {
    std::lock_guard<std::mutex> lock(pool().mu);
    auto& v = pool().idle;
    for (size_t i = 0; i < v.size(); i++)
      if (v[i]->device == device) {
        w = v[i];
        v.erase(v.begin() + (long)i);
        break;
      }
  }
  if (hipSetDevice(device) != hipSuccess) {
    er::fail("hipSetDevice(%d) failed", device);
    return nullptr;
  }
  if (!w) {
    w = new IcpWs();
    w->device = device;
    if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&w->ev, hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void**)&w->icount, 4 * sizeof(int)) != hipSuccess || hipMalloc((void**)&w->acc, kAcc * sizeof(double)) != hipSuccess ||
        hipMalloc((void**)&w->dstate, sizeof(IcpDev)) != hipSuccess ||
        hipHostMalloc((void**)&w->host, sizeof(HostBlock), hipHostMallocDefault) != hipSuccess) {
      er::fail("ICP workspace allocation failed: %s", hipGetErrorString(hipGetLastError()));
      ws_destroy(w);
      return nullptr;
    }
  }
  if (ws_reserve(w, n)) {
    ws_destroy(w);
    return nullptr;
  }
  return w;
}

void ws_release(IcpWs* w) {
  if (!w) return;
  std::lock_guard<std::mutex> lock(pool().mu);
  pool().idle.push_back(w);
}

struct WsSet {                  // RAII: the workspaces of one API call
  std::vector<IcpWs*> ws;
  ~WsSet() {
    for (IcpWs* w : ws) {
      if (w->stream) (void)hipStreamSynchronize(w->stream);
      ws_release(w);
    }
  }
  int acquire(int device, size_t n, int count) {
    for (int i = 0; i < count; i++) {
      IcpWs* w = ws_acquire(device, n);
      if (!w) return 1;
      ws.push_back(w);
    }
    return 0;
  }
};

static int lanes_cfg() { const char* e = getenv("ER_ICP_LANES"); int v = e ? atoi(e) : 4; return v < 1 ? 1 : (v > 16 ? 16 : v); }
#define kLanes (lanes_cfg())

int nblocks_of(int n) { return (std::max(n, 1) + kBlock - 1) / kBlock; }
int gblocks_of(int n) { return nblocks_of(n); }                // NN kernels: one query per thread

int check_pair(er_cloud_t src, er_cloud_t tgt, double radius, const char* who) {
  if (!src || !tgt) return er::fail("%s: NULL cloud", who);
  if (src->device != tgt->device) return er::fail("%s: source and target live on different devices", who);
  if (!(radius > 0.0) || radius > (double)tgt->radius_cap * (1.0 + 1e-6))
    return er::fail("%s: search radius %g exceeds the target's grid cell %g (er_cloud_create grid_cell)", who, radius, (double)tgt->radius_cap);
  return 0;
}

// ---- Registration pre-check ---------------------------------------------------------------------
int count_enqueue(IcpWs* w, er_cloud_t src, er_cloud_t tgt, const double* T, double max_dist) {
  Mat12d M;
  for (int q = 0; q < 12; q++) M.m[q] = T[q];
  ER_HIP_TRY(hipMemsetAsync(w->icount, 0, sizeof(int), w->stream));
  if (src->n > 0 && tgt->n > 0) {
    hipLaunchKernelGGL(k_count_inliers, dim3(std::min(gblocks_of(src->n), 2048)), dim3(kBlock), 0, w->stream, src->sorted, src->n, M,
                       grid_of(tgt), (float)max_dist, max_dist * max_dist, w->icount);
    ER_HIP_TRY(hipGetLastError());
  }
  ER_HIP_TRY(hipMemcpyAsync(&w->host->count[0], w->icount, sizeof(int), hipMemcpyDeviceToHost, w->stream));
  ER_HIP_TRY(hipEventRecord(w->ev, w->stream));
  return 0;
}

// ---- ICP as a resumable job ---------------------------------------------------------------------
struct AlignJob {
  er_cloud_t src = nullptr, tgt = nullptr;
  float fin[16];
  int iter = 0;
  bool conv = false;
  double fitness = DBL_MAX;
  enum { ITERATING, FITNESS, DONE } state = ITERATING;
};

struct AlignParams {
  double max_dist, eps;
  int max_iter, stop_rule;
  bool want_fitness;
};

// Iterations enqueued per host visit.  PCL's loop runs 3 iterations on most fragment pairs of the pipeline (the third one
// meets the stop rule): one chunk usually ends the job; launches of a chunk that come after the stop decision return at once.
constexpr int kIcpChunk = 3;

// A chunk of ICP iterations with NO host round trip in between: k_icp_iter (apply the last increment, exact NN, point-to-plane
// sums) + k_icp_final (fixed-order total, 6x6 solve, increment, stop rule -- on the device), then the state comes back once.
int align_enqueue_chunk(IcpWs* w, AlignJob& j, const AlignParams& P) {
  const int n = j.src->n;
  const IcpParams IP{P.eps, P.max_iter, P.stop_rule};
  for (int c = 0; c < kIcpChunk; c++) {
    hipLaunchKernelGGL(k_icp_iter, dim3(nblocks_of(n)), dim3(kBlock), 0, w->stream, w->X, n, w->dstate, grid_of(j.tgt), (float)P.max_dist,
                       P.max_dist * P.max_dist, j.tgt->xyz, j.tgt->nrm, w->partial);
    hipLaunchKernelGGL(k_icp_final, dim3(1), dim3(kBlock), 0, w->stream, w->partial, nblocks_of(n), w->acc, w->dstate, IP);
  }
  ER_HIP_TRY(hipGetLastError());
  ER_HIP_TRY(hipMemcpyAsync(&w->host->state, w->dstate, sizeof(IcpDev), hipMemcpyDeviceToHost, w->stream));
  ER_HIP_TRY(hipEventRecord(w->ev, w->stream));
  return 0;
}

int align_enqueue_fitness(IcpWs* w, AlignJob& j, const AlignParams& P) {
  const int n = j.src->n;
  Mat12f F;
  for (int q = 0; q < 12; q++) F.m[q] = j.fin[q];
  ER_HIP_TRY(hipMemsetAsync(w->acc, 0, kAcc * sizeof(double), w->stream));
  if (n > 0 && j.tgt->n > 0) {
    hipLaunchKernelGGL(k_fitness_nn, dim3(gblocks_of(n)), dim3(kBlock), 0, w->stream, j.src->sorted, n, F, grid_of(j.tgt), (float)P.max_dist, w->nd);
    hipLaunchKernelGGL(k_fitness_sum, dim3(nblocks_of(n)), dim3(kBlock), 0, w->stream, w->nd, n, w->acc);
    ER_HIP_TRY(hipGetLastError());
  }
  ER_HIP_TRY(hipMemcpyAsync(w->host->acc, w->acc, 2 * sizeof(double), hipMemcpyDeviceToHost, w->stream));
  ER_HIP_TRY(hipEventRecord(w->ev, w->stream));
  j.state = AlignJob::FITNESS;
  return 0;
}

int align_start(IcpWs* w, AlignJob& j, er_cloud_t src, er_cloud_t tgt, const float* guess, const AlignParams& P) {
  j = AlignJob();
  j.src = src;
  j.tgt = tgt;
  memcpy(j.fin, guess, sizeof j.fin);                        // final_transformation_ = guess
  if (src->n == 0 || tgt->n == 0) {                          // fewer than 3 correspondences by construction: nothing to enqueue
                                                             // (max_iter <= 0 still runs ONE iteration, like PCL's do { } while loop)
    j.iter = 0;
    j.conv = false;
    if (P.want_fitness) return align_enqueue_fitness(w, j, P);
    j.state = AlignJob::DONE;
    ER_HIP_TRY(hipEventRecord(w->ev, w->stream));
    return 0;
  }
  bool ident = true;
  for (int i = 0; i < 16; i++) ident = ident && guess[i] == ((i % 5 == 0) ? 1.f : 0.f);
  // The previous job's state read-back has been consumed (the lane's event was waited for), so the pinned block is free.
  IcpDev& st = w->host->state;
  memset(&st, 0, sizeof st);
  memcpy(st.fin, guess, sizeof st.fin);
  for (int i = 0; i < 16; i++) st.delta[i] = st.prev_delta[i] = (i % 5 == 0) ? 1.f : 0.f;
  st.prev_mse = DBL_MAX;
  ER_HIP_TRY(hipMemcpyAsync(w->dstate, &st, sizeof st, hipMemcpyHostToDevice, w->stream));
  Mat12f G;
  for (int q = 0; q < 12; q++) G.m[q] = guess[q];
  hipLaunchKernelGGL(k_init_x, dim3(nblocks_of(src->n)), dim3(kBlock), 0, w->stream, src->sorted, w->X, src->n, G, ident ? 0 : 1);
  ER_HIP_TRY(hipGetLastError());
  return align_enqueue_chunk(w, j, P);
}

// The host half of a job: wait for the lane's event; either the loop has ended on the device (collect) or another chunk goes out.
int align_advance(IcpWs* w, AlignJob& j, const AlignParams& P) {
  ER_HIP_TRY(hipEventSynchronize(w->ev));
  if (j.state == AlignJob::DONE) return 0;
  if (j.state == AlignJob::FITNESS) {
    const double* acc = w->host->acc;
    j.fitness = acc[1] > 0 ? acc[0] / acc[1] : DBL_MAX;
    j.state = AlignJob::DONE;
    return 0;
  }
  const IcpDev& st = w->host->state;
  if (!st.done) return align_enqueue_chunk(w, j, P);         // (h2d of the state is NOT repeated: it lives on the device)
  memcpy(j.fin, st.fin, sizeof j.fin);
  j.iter = st.iter;
  j.conv = st.conv != 0;
  if (P.want_fitness) return align_enqueue_fitness(w, j, P);
  j.state = AlignJob::DONE;
  return 0;
}

// ---- FindCorrespondence ---------------------------------------------------------------------------
int corr_enqueue(IcpWs* w, er_cloud_t src, er_cloud_t tgt, const double* T, double dist, double normal_cos, bool want_info) {
  const int n = src->n;
  Mat12d M;
  for (int q = 0; q < 12; q++) M.m[q] = T[q];
  const int nb = nblocks_of(n);
  ER_HIP_TRY(hipMemsetAsync(w->acc, 0, kAcc * sizeof(double), w->stream));
  hipLaunchKernelGGL(k_find_corr, dim3(gblocks_of(n)), dim3(kBlock), 0, w->stream, src->sorted, src->nrm, n, M, grid_of(tgt), tgt->nrm,
                     (float)dist, dist * dist, normal_cos, w->match);
  hipLaunchKernelGGL(k_count_blocks, dim3(nb), dim3(kBlock), 0, w->stream, w->match, src->xyz, n, w->block_count, w->acc, want_info ? 1 : 0);
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, w->stream, w->block_count, w->block_offset, nb, w->icount + 1);
  hipLaunchKernelGGL(k_compact, dim3(nb), dim3(kBlock), 0, w->stream, w->match, n, w->block_offset, w->pairs, n);
  ER_HIP_TRY(hipGetLastError());
  ER_HIP_TRY(hipMemcpyAsync(&w->host->count[1], w->icount + 1, sizeof(int), hipMemcpyDeviceToHost, w->stream));
  ER_HIP_TRY(hipMemcpyAsync(w->host->acc, w->acc, 10 * sizeof(double), hipMemcpyDeviceToHost, w->stream));
  ER_HIP_TRY(hipEventRecord(w->ev, w->stream));
  return 0;
}

bool is_pinned_host(const void* p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();                                 // plain malloc memory: "invalid value", not an error for us
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

// sum A^T A with A = [I | B], B = [[0, 2sz, -2sy], [-2sz, 0, 2sx], [2sy, -2sx, 0]]  (CorresApp.cpp:198-203,
// RansacCurvature.h:717-721) from its ten distinct terms (k_count_blocks / k_info_matched).
void expand_information(const double* acc, double* I) {
  memset(I, 0, 36 * sizeof(double));
  const double N = acc[9];
  I[0 * 6 + 0] = I[1 * 6 + 1] = I[2 * 6 + 2] = N;
  I[0 * 6 + 4] = I[4 * 6 + 0] = acc[2];      //  sum 2sz
  I[0 * 6 + 5] = I[5 * 6 + 0] = -acc[1];     // -sum 2sy
  I[1 * 6 + 3] = I[3 * 6 + 1] = -acc[2];
  I[1 * 6 + 5] = I[5 * 6 + 1] = acc[0];      //  sum 2sx
  I[2 * 6 + 3] = I[3 * 6 + 2] = acc[1];
  I[2 * 6 + 4] = I[4 * 6 + 2] = -acc[0];
  I[3 * 6 + 3] = acc[3];
  I[4 * 6 + 4] = acc[4];
  I[5 * 6 + 5] = acc[5];
  I[3 * 6 + 4] = I[4 * 6 + 3] = acc[6];
  I[3 * 6 + 5] = I[5 * 6 + 3] = acc[7];
  I[4 * 6 + 5] = I[5 * 6 + 4] = acc[8];
}

// (Letting k_compact store the list straight into page-locked host memory -- zero-copy -- was tried: one stage
// fewer, but 3.5 instead of 2.9 ms per 40 pairs.)
// Stage 2 of a pair (after the lane's kernel event): the pair count is known; start the copy of exactly that many
// pairs -- straight into the caller's buffer when it is page-locked (er_host_alloc), else into the lane's pinned
// staging block -- and expand the information matrix.  *staged tells corr_finish whether a host memcpy remains.
int corr_start_copy(IcpWs* w, int* pairs_host, int capacity, int* n_pairs, double* info36, bool* staged) {
  ER_HIP_TRY(hipEventSynchronize(w->ev));
  const int total = w->host->count[1];
  *n_pairs = total;
  *staged = false;
  const int ncopy = std::min(total, capacity);
  if (ncopy > 0) {
    int* dst = pairs_host;
    if (!is_pinned_host(pairs_host)) {
      if (w->stage_cap < w->cap) {
        if (w->stage) (void)hipHostFree(w->stage);
        w->stage = nullptr;
        w->stage_cap = 0;
        ER_HIP_TRY(hipHostMalloc((void**)&w->stage, w->cap * 2 * sizeof(int), hipHostMallocDefault));
        w->stage_cap = w->cap;
      }
      dst = w->stage;
      *staged = true;
    }
    ER_HIP_TRY(hipMemcpyAsync(dst, w->pairs, (size_t)ncopy * 2 * sizeof(int), hipMemcpyDeviceToHost, w->stream));
    ER_HIP_TRY(hipEventRecord(w->ev, w->stream));
  }
  if (info36) expand_information(w->host->acc, info36);
  return 0;
}

// Stage 3: the copy has landed.
int corr_finish(IcpWs* w, int* pairs_host, int capacity, int total, bool staged) {
  const int ncopy = std::min(total, capacity);
  if (ncopy > 0) {
    ER_HIP_TRY(hipEventSynchronize(w->ev));
    if (staged) memcpy(pairs_host, w->stage, (size_t)ncopy * 2 * sizeof(int));
  }
  return total > capacity ? er::fail("er_find_correspondence: %d pairs exceed the capacity %d", total, capacity) : 0;
}

int batch_prologue(int n, const er_cloud_t* src, const er_cloud_t* tgt, double radius, const char* who, int* device, size_t* max_n) {
  if (n < 0 || (n > 0 && (!src || !tgt))) return er::fail("%s: bad arguments", who);
  *device = n > 0 && src[0] ? src[0]->device : 0;
  *max_n = 1;
  for (int i = 0; i < n; i++) {
    if (check_pair(src[i], tgt[i], radius, who)) return 1;
    if (src[i]->device != *device) return er::fail("%s: all pairs of one batch must live on one device", who);
    *max_n = std::max(*max_n, (size_t)src[i]->n);
  }
  return 0;
}

}  // namespace

extern "C" {

int er_cloud_create(const float* xyz_host, const float* normal_host, int n, float grid_cell, int device, er_cloud_t* out) {
  if (!out) return er::fail("er_cloud_create: out is NULL");
  *out = nullptr;
  if (n < 0 || (n > 0 && (!xyz_host || !normal_host))) return er::fail("er_cloud_create: bad arguments");
  if (!(grid_cell > 0.f)) return er::fail("er_cloud_create: grid_cell must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return er::fail("er_cloud_create: no HIP device available (liber_hip has no CPU fallback)");
  if (device < 0 || device >= ndev) return er::fail("er_cloud_create: device %d out of range [0,%d)", device, ndev);
  ER_HIP_TRY(hipSetDevice(device));
  er_cloud_t c = new er_cloud_s();
  c->device = device;
  c->n = n;
  c->radius_cap = grid_cell;
  const size_t nn = (size_t)std::max(n, 1);
#define ER_CALLOC(ptr, bytes)                                                                 \
  do {                                                                                        \
    hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                                       \
    if (e_ != hipSuccess) {                                                                   \
      er::fail("er_cloud_create: hipMalloc(%zu) failed: %s", (size_t)(bytes), hipGetErrorString(e_)); \
      er_cloud_destroy(c);                                                                    \
      return 1;                                                                               \
    }                                                                                         \
  } while (0)
#define ER_CTRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      er::fail("er_cloud_create: %s failed: %s", #expr, hipGetErrorString(e_));               \
      er_cloud_destroy(c);                                                                    \
      return 1;                                                                               \
    }                                                                                         \
  } while (0)
  // one allocation for the three per-point arrays: [xyz 3n | normals 3n | sorted n float4]
  const size_t off_sorted = (nn * 6 + 3) / 4 * 4;                 // float4 needs 16-byte alignment
  ER_CALLOC(c->xyz, (off_sorted + nn * 4) * sizeof(float));
  c->nrm = c->xyz + nn * 3;
  c->sorted = reinterpret_cast<float4*>(c->xyz + off_sorted);
  if (n > 0) {
    ER_CTRY(hipMemcpyAsync(c->xyz, xyz_host, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, nullptr));
    ER_CTRY(hipMemcpyAsync(c->nrm, normal_host, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, nullptr));
  }
  // ---- uniform grid on the DEVICE (once per fragment; every pair that uses it as target reuses it) ----
  float cell = grid_cell * 1.001f;               // strictly larger than any admissible radius
  float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  int dim[3] = {1, 1, 1};
  GridScratch& gs = grid_scratch(device);
  std::lock_guard<std::mutex> lock(gs.mu);
  if (!gs.bounds) ER_CTRY(hipMalloc((void**)&gs.bounds, 8 * sizeof(int)));
  if (n > 0) {
    const int init[8] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN, 0, 0};
    int got[8];
    ER_CTRY(hipMemcpyAsync(gs.bounds, init, sizeof init, hipMemcpyHostToDevice, nullptr));
    hipLaunchKernelGGL(k_grid_bounds, dim3(std::min(nblocks_of(n), 128)), dim3(kBlock), 0, nullptr, c->xyz, n, gs.bounds);
    ER_CTRY(hipMemcpyAsync(got, gs.bounds, sizeof got, hipMemcpyDeviceToHost, nullptr));
    ER_CTRY(hipStreamSynchronize(nullptr));
    if (got[6]) {
      er_cloud_destroy(c);
      return er::fail("er_cloud_create: non-finite coordinates");
    }
    for (int a = 0; a < 3; a++) {
      lo[a] = ordered_float(got[a]);
      hi[a] = ordered_float(got[3 + a]);
    }
  }
  for (;;) {
    long total = 1;
    for (int a = 0; a < 3; a++) {
      dim[a] = (int)std::floor((hi[a] - lo[a]) / cell) + 1;
      total *= dim[a];
    }
    if (total <= (1L << 25)) break;
    cell *= 2.f;
  }
  const int ncell = dim[0] * dim[1] * dim[2];
  ER_CALLOC(c->cell_start, ((size_t)ncell + 1) * sizeof(int));
  ER_CTRY(hipMemsetAsync(c->cell_start, 0, ((size_t)ncell + 1) * sizeof(int), nullptr));
  if (n > 0) {
    if (gs.n_cap < (size_t)n) {
      for (int q = 0; q < 2; q++) {
        if (gs.key[q]) (void)hipFree(gs.key[q]);
        if (gs.idx[q]) (void)hipFree(gs.idx[q]);
        gs.key[q] = gs.idx[q] = nullptr;
      }
      gs.n_cap = 0;
      const size_t cap = (size_t)n + (size_t)n / 8;
      for (int q = 0; q < 2; q++) {
        ER_CTRY(hipMalloc((void**)&gs.key[q], cap * sizeof(unsigned)));
        ER_CTRY(hipMalloc((void**)&gs.idx[q], cap * sizeof(unsigned)));
      }
      gs.n_cap = cap;
    }
    GridDims G;
    for (int a = 0; a < 3; a++) {
      G.org[a] = lo[a];
      G.dim[a] = dim[a];
    }
    G.cell = cell;
    hipLaunchKernelGGL(k_grid_cells, dim3(nblocks_of(n)), dim3(kBlock), 0, nullptr, c->xyz, n, G, gs.key[0], gs.idx[0], c->cell_start);
    int bits = 1;
    while ((1L << bits) < (long)ncell) bits++;
    size_t need_sort = 0, need_scan = 0;
    ER_CTRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need_sort, gs.key[0], gs.key[1], gs.idx[0], gs.idx[1], n, 0, bits, (hipStream_t) nullptr));
    ER_CTRY(hipcub::DeviceScan::InclusiveSum(nullptr, need_scan, c->cell_start, c->cell_start, ncell + 1, (hipStream_t) nullptr));
    const size_t need = std::max(need_sort, need_scan);
    if (gs.cub_cap < need) {
      if (gs.cub) (void)hipFree(gs.cub);
      gs.cub = nullptr;
      gs.cub_cap = 0;
      ER_CTRY(hipMalloc(&gs.cub, need + need / 4));
      gs.cub_cap = need + need / 4;
    }
    size_t tmp = gs.cub_cap;
    ER_CTRY(hipcub::DeviceRadixSort::SortPairs(gs.cub, tmp, gs.key[0], gs.key[1], gs.idx[0], gs.idx[1], n, 0, bits, (hipStream_t) nullptr));
    tmp = gs.cub_cap;
    ER_CTRY(hipcub::DeviceScan::InclusiveSum(gs.cub, tmp, c->cell_start, c->cell_start, ncell + 1, (hipStream_t) nullptr));
    hipLaunchKernelGGL(k_grid_gather, dim3(nblocks_of(n)), dim3(kBlock), 0, nullptr, c->xyz, gs.idx[1], n, c->sorted);
    ER_CTRY(hipGetLastError());
  }
  ER_CTRY(hipStreamSynchronize(nullptr));          // the caller's host arrays and the shared scratch are free again
#undef ER_CALLOC
#undef ER_CTRY
  c->grid.pts = c->sorted;
  c->grid.cell_start = c->cell_start;
  c->grid.cell = cell;
  for (int a = 0; a < 3; a++) {
    c->grid.org[a] = lo[a];
    c->grid.dim[a] = dim[a];
  }
  *out = c;
  return 0;
}

int er_cloud_destroy(er_cloud_t c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  void* ptrs[] = {c->xyz, c->cell_start};          // xyz heads the one allocation that also holds nrm and sorted
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete c;
  return 0;
}

int er_cloud_size(er_cloud_t c) { return c ? c->n : -1; }

int er_icp_release_workspaces(void) {
  std::vector<IcpWs*> all;
  {
    std::lock_guard<std::mutex> lock(pool().mu);
    all.swap(pool().idle);
  }
  for (IcpWs* w : all) ws_destroy(w);
  return 0;
}

int er_icp_count_inliers_batch(int n, const er_cloud_t* src, const er_cloud_t* tgt, const double* T, double max_dist, int* counts) {
  int device;
  size_t max_n;
  if (n > 0 && (!T || !counts)) return er::fail("er_icp_count_inliers_batch: NULL argument");
  if (batch_prologue(n, src, tgt, max_dist, "er_icp_count_inliers", &device, &max_n)) return 1;
  if (n == 0) return 0;
  WsSet set;
  const int lanes = std::min(n, kLanes);
  if (set.acquire(device, 1, lanes)) return 1;               // the pre-check needs no per-point scratch
  for (int i = 0; i < n + lanes; i++) {
    IcpWs* w = set.ws[(size_t)(i % lanes)];
    if (i >= lanes) {
      ER_HIP_TRY(hipEventSynchronize(w->ev));
      counts[i - lanes] = w->host->count[0];
    }
    if (i < n && count_enqueue(w, src[i], tgt[i], T + (size_t)i * 16, max_dist)) return 1;
  }
  return 0;
}

int er_icp_count_inliers(er_cloud_t src, er_cloud_t tgt, const double T[16], double max_dist, int* count) {
  if (!T || !count) return er::fail("er_icp_count_inliers: NULL argument");
  return er_icp_count_inliers_batch(1, &src, &tgt, T, max_dist, count);
}

int er_icp_align_batch(int n, const er_cloud_t* src, const er_cloud_t* tgt, const float* guess, double max_dist, int max_iter,
                       double transformation_epsilon, int stop_rule, float* out, int* iterations, int* converged, double* fitness) {
  int device;
  size_t max_n;
  if (n > 0 && (!guess || !out)) return er::fail("er_icp_align: NULL argument");
  if (batch_prologue(n, src, tgt, max_dist, "er_icp_align", &device, &max_n)) return 1;
  if (n == 0) return 0;
  const AlignParams P{max_dist, transformation_epsilon, max_iter, stop_rule, fitness != nullptr};
  WsSet set;
  const int lanes = std::min(n, kLanes);
  if (set.acquire(device, max_n, lanes)) return 1;
  std::vector<AlignJob> job((size_t)lanes);
  std::vector<int> which((size_t)lanes, -1);
  int next = 0, done = 0;
  for (int l = 0; l < lanes; l++) {
    which[(size_t)l] = next;
    if (align_start(set.ws[(size_t)l], job[(size_t)l], src[next], tgt[next], guess + (size_t)next * 16, P)) return 1;
    next++;
  }
  // round robin over the lanes: every lane always has work queued, so waiting on one never idles the GPU
  for (int l = 0; done < n; l = (l + 1) % lanes) {
    const int i = which[(size_t)l];
    if (i < 0) continue;
    AlignJob& j = job[(size_t)l];
    if (align_advance(set.ws[(size_t)l], j, P)) return 1;
    if (j.state != AlignJob::DONE) continue;
    memcpy(out + (size_t)i * 16, j.fin, sizeof j.fin);
    if (iterations) iterations[i] = j.iter;
    if (converged) converged[i] = j.conv ? 1 : 0;
    if (fitness) fitness[i] = j.fitness;
    done++;
    which[(size_t)l] = -1;
    if (next < n) {
      which[(size_t)l] = next;
      if (align_start(set.ws[(size_t)l], j, src[next], tgt[next], guess + (size_t)next * 16, P)) return 1;
      next++;
    }
  }
  return 0;
}

int er_icp_align(er_cloud_t src, er_cloud_t tgt, const float guess[16], double max_dist, int max_iter,
                 double transformation_epsilon, int stop_rule, float out[16], int* iterations, int* converged,
                 double* fitness) {
  if (!guess || !out) return er::fail("er_icp_align: NULL argument");
  return er_icp_align_batch(1, &src, &tgt, guess, max_dist, max_iter, transformation_epsilon, stop_rule, out, iterations, converged,
                            fitness);
}

int er_ransac_fitness_batch(er_cloud_t src, er_cloud_t tgt, int n_hyp, const float* M, float corr_dist_threshold, int* inliers,
                            double* fitness) {
  if (n_hyp < 0 || (n_hyp > 0 && (!M || !inliers))) return er::fail("er_ransac_fitness_batch: bad arguments");
  if (check_pair(src, tgt, (double)corr_dist_threshold, "er_ransac_fitness_batch")) return 1;
  if (n_hyp == 0) return 0;
  WsSet set;
  if (set.acquire(src->device, 1, 1)) return 1;
  IcpWs* w = set.ws[0];
  float* d_hyp = nullptr;
  int* d_cnt = nullptr;
  double* d_sum = nullptr;
  std::vector<double> sums((size_t)n_hyp, 0.0);
  int rc = 0;
  const size_t nh = (size_t)n_hyp;
  if (hipMalloc((void**)&d_hyp, nh * 16 * sizeof(float)) != hipSuccess || hipMalloc((void**)&d_cnt, nh * sizeof(int)) != hipSuccess ||
      hipMalloc((void**)&d_sum, nh * sizeof(double)) != hipSuccess) {
    rc = er::fail("er_ransac_fitness_batch: hipMalloc failed: %s", hipGetErrorString(hipGetLastError()));
  } else if (hipMemcpyAsync(d_hyp, M, nh * 16 * sizeof(float), hipMemcpyHostToDevice, w->stream) != hipSuccess ||
             hipMemsetAsync(d_cnt, 0, nh * sizeof(int), w->stream) != hipSuccess ||
             hipMemsetAsync(d_sum, 0, nh * sizeof(double), w->stream) != hipSuccess) {
    rc = er::fail("er_ransac_fitness_batch: upload failed: %s", hipGetErrorString(hipGetLastError()));
  } else {
    if (src->n > 0 && tgt->n > 0) {
      // few workgroups per hypothesis when there are many hypotheses (the reference's clouds are down-sampled), all of them otherwise
      const int bx = std::max(1, std::min(nblocks_of(src->n), 4096 / std::min(n_hyp, 4096)));
      for (int h0 = 0; h0 < n_hyp && rc == 0; h0 += 32768) {
        const int hn = std::min(32768, n_hyp - h0);
        hipLaunchKernelGGL(k_ransac_fitness, dim3(bx, hn), dim3(kBlock), 0, w->stream, src->sorted, src->n, d_hyp, h0, grid_of(tgt),
                           corr_dist_threshold, corr_dist_threshold * corr_dist_threshold, d_cnt, d_sum);
        if (hipGetLastError() != hipSuccess) rc = er::fail("er_ransac_fitness_batch: launch failed");
      }
    }
    if (rc == 0 && (hipMemcpyAsync(inliers, d_cnt, nh * sizeof(int), hipMemcpyDeviceToHost, w->stream) != hipSuccess ||
                    hipMemcpyAsync(sums.data(), d_sum, nh * sizeof(double), hipMemcpyDeviceToHost, w->stream) != hipSuccess ||
                    hipStreamSynchronize(w->stream) != hipSuccess))
      rc = er::fail("er_ransac_fitness_batch: %s", hipGetErrorString(hipGetLastError()));
  }
  if (d_hyp) (void)hipFree(d_hyp);
  if (d_cnt) (void)hipFree(d_cnt);
  if (d_sum) (void)hipFree(d_sum);
  if (rc == 0 && fitness)
    for (int h = 0; h < n_hyp; h++) fitness[h] = inliers[h] > 0 ? sums[(size_t)h] / (double)inliers[h] : (double)FLT_MAX;   // :697-703
  return rc;
}

int er_ransac_inliers(er_cloud_t src, er_cloud_t tgt, const float* M16, float corr_dist_threshold, int* pairs_host, int capacity,
                      int* n_inliers, double* fitness, double* info_source36, double* info_target36) {
  if (!M16 || !n_inliers || capacity < 0 || (capacity > 0 && !pairs_host)) return er::fail("er_ransac_inliers: bad arguments");
  if (check_pair(src, tgt, (double)corr_dist_threshold, "er_ransac_inliers")) return 1;
  *n_inliers = 0;
  if (fitness) *fitness = (double)FLT_MAX;
  if (info_source36) memset(info_source36, 0, 36 * sizeof(double));
  if (info_target36) memset(info_target36, 0, 36 * sizeof(double));
  if (src->n == 0 || tgt->n == 0) return 0;
  WsSet set;
  if (set.acquire(src->device, (size_t)src->n, 1)) return 1;
  IcpWs* w = set.ws[0];
  const int n = src->n, nb = nblocks_of(n);
  Mat12f M;
  for (int q = 0; q < 12; q++) M.m[q] = M16[q];
  ER_HIP_TRY(hipMemsetAsync(w->acc, 0, kAcc * sizeof(double), w->stream));
  hipLaunchKernelGGL(k_ransac_match, dim3(gblocks_of(n)), dim3(kBlock), 0, w->stream, src->sorted, n, M, grid_of(tgt), corr_dist_threshold,
                     corr_dist_threshold * corr_dist_threshold, w->match, w->acc);
  hipLaunchKernelGGL(k_count_blocks, dim3(nb), dim3(kBlock), 0, w->stream, w->match, src->xyz, n, w->block_count, w->acc, info_source36 ? 1 : 0);
  if (info_target36) hipLaunchKernelGGL(k_info_matched, dim3(nb), dim3(kBlock), 0, w->stream, w->match, tgt->xyz, n, w->acc + 10);
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, w->stream, w->block_count, w->block_offset, nb, w->icount + 1);
  hipLaunchKernelGGL(k_compact, dim3(nb), dim3(kBlock), 0, w->stream, w->match, n, w->block_offset, w->pairs, n);
  ER_HIP_TRY(hipGetLastError());
  ER_HIP_TRY(hipMemcpyAsync(&w->host->count[1], w->icount + 1, sizeof(int), hipMemcpyDeviceToHost, w->stream));
  ER_HIP_TRY(hipMemcpyAsync(w->host->acc, w->acc, kAcc * sizeof(double), hipMemcpyDeviceToHost, w->stream));
  ER_HIP_TRY(hipEventRecord(w->ev, w->stream));
  bool staged = false;
  if (corr_start_copy(w, pairs_host, capacity, n_inliers, info_source36, &staged)) return 1;
  if (info_target36) expand_information(w->host->acc + 10, info_target36);
  if (fitness && *n_inliers > 0) *fitness = w->host->acc[20] / (double)*n_inliers;                  // :697-703
  return corr_finish(w, pairs_host, capacity, *n_inliers, staged);
}

int er_find_correspondence_batch(int n, const er_cloud_t* src, const er_cloud_t* tgt, const double* T, double dist, double normal_cos,
                                 int* const* pairs_host, const int* capacity, int* n_pairs, double* info36) {
  int device;
  size_t max_n;
  if (n > 0 && (!T || !n_pairs || !pairs_host || !capacity)) return er::fail("er_find_correspondence: NULL argument");
  if (batch_prologue(n, src, tgt, dist, "er_find_correspondence", &device, &max_n)) return 1;
  if (n == 0) return 0;
  for (int i = 0; i < n; i++)
    if (capacity[i] > 0 && !pairs_host[i]) return er::fail("er_find_correspondence: NULL pair buffer for pair %d", i);
  WsSet set;
  const int lanes = std::min(n, kLanes);
  if (set.acquire(device, max_n, lanes)) return 1;
  // per lane: IDLE -> KERNELS (search + compaction queued) -> COPY (pair list on its way to the host) -> IDLE
  enum { IDLE, KERNELS, COPY };
  struct Lane { int state = IDLE, pair = -1; bool staged = false; };
  std::vector<Lane> lane((size_t)lanes);
  int next = 0, done = 0, rc = 0;
  for (int l = 0; done < n; l = (l + 1) % lanes) {
    Lane& L = lane[(size_t)l];
    IcpWs* w = set.ws[(size_t)l];
    if (L.state == COPY) {
      if (corr_finish(w, pairs_host[L.pair], capacity[L.pair], n_pairs[L.pair], L.staged)) rc = 1;   // keep draining the other lanes
      L.state = IDLE;
      done++;
    } else if (L.state == KERNELS) {
      const int k = L.pair;
      if (corr_start_copy(w, pairs_host[k], capacity[k], &n_pairs[k], info36 ? info36 + (size_t)k * 36 : nullptr, &L.staged)) return 1;
      L.state = COPY;
      continue;                                              // the lane's buffers stay busy until the copy is collected
    }
    if (L.state == IDLE && next < n) {
      const int k = next++;
      if (src[k]->n == 0 || tgt[k]->n == 0) {
        n_pairs[k] = 0;
        if (info36) memset(info36 + (size_t)k * 36, 0, 36 * sizeof(double));
        done++;
        continue;
      }
      if (corr_enqueue(w, src[k], tgt[k], T + (size_t)k * 16, dist, normal_cos, info36 != nullptr)) return 1;
      L.pair = k;
      L.state = KERNELS;
    }
  }
  return rc;
}

int er_find_correspondence(er_cloud_t src, er_cloud_t tgt, const double T[16], double dist, double normal_cos,
                           int* pairs_host, int capacity, int* n_pairs, double* info36) {
  if (!T || !n_pairs || (capacity > 0 && !pairs_host)) return er::fail("er_find_correspondence: NULL argument");
  int* const bufs[1] = {pairs_host};
  return er_find_correspondence_batch(1, &src, &tgt, T, dist, normal_cos, bufs, &capacity, n_pairs, info36);
}

}  // extern "C"
