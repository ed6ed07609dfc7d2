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
//   X, match...  per-cloud scratch used when the cloud is the SOURCE of a pair
// Queries run in the SOURCE cloud's own cell-sorted order (thread t takes sorted[t]), so the lanes of a wave
// walk the same few target cells together (coalesced / broadcast candidate loads); results are written back
// by original index.  NN kernels use 8 lanes per source point (each lane scans part of the 9 cell rows, shuffle-min);
// candidates stream from L2 as 16-byte loads:
//   k_count_inliers   transform (float64 -> float32) + NN + count           (Registration pre-check)
//   k_icp_nn          [apply last increment] + NN -> nn[], d2[];  k_icp_accum: point-to-plane rows -> 27+2 float64
//                     sums (wave shuffle -> LDS -> one atomicAdd per block and sum)
//   k_find_corr       transform points+normals + NN + distance/normal tests -> match[orig index];
//                     k_count_blocks (+ information-matrix sums) + k_scan_blocks + k_compact = stable compaction in file order
// Reductions and scans, not contractions: no MFMA.
#include "er_common.h"

#include "../../include/er_hip.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
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

// Exact 1-NN of q among target points inside the 27 neighbouring cells, searched by a group of kGroup
// consecutive lanes per query: lane `sub` scans the rows of cells sub, sub + kGroup, ... of the 9 (z,y) rows
// (home row first; rows that cannot beat the best so far are skipped), then the group reduces (distance, index)
// lexicographically with shuffles.  The pass is latency bound (dependent cell_start -> candidate loads).
// float32 squared distance ((dx*dx) + dy*dy) + dz*dz (FLANN L2_Simple), ties -> lower original index.
// limit2 = squared search radius: callers discard anything farther, so rows of cells lying entirely beyond
// the radius are skipped (margin 1e-4 relative for the float32 cell assignment).  All lanes of the group
// return the same (index or -1, distance).
#ifndef ER_ICP_GROUP
#define ER_ICP_GROUP 2      /* measured on MI355X: 1/2/4/8 lanes per query -> NN pass 58/51/57/64 us per 253 k queries */
#endif
constexpr int kGroup = ER_ICP_GROUP;

__device__ __forceinline__ int nn_search(const Grid& g, float qx, float qy, float qz, float limit2, int sub, float& best_d) {
  const float ux = (qx - g.org[0]) / g.cell, uy = (qy - g.org[1]) / g.cell, uz = (qz - g.org[2]) / g.cell;
  const float cx = floorf(ux), cy = floorf(uy), cz = floorf(uz);
  int best = -1;
  float bd = FLT_MAX;
  const bool inside = cx >= -1.f && cx <= (float)g.dim[0] && cy >= -1.f && cy <= (float)g.dim[1] && cz >= -1.f && cz <= (float)g.dim[2];
  if (inside) {
    const int ix = (int)cx, iy = (int)cy, iz = (int)cz;
    const int x0 = max(ix - 1, 0), x1 = min(ix + 1, g.dim[0] - 1);
    // distance from q to the lower / upper face of its own cell along y and z (metres)
    const float ylo = (uy - cy) * g.cell, yhi = g.cell - ylo, zlo = (uz - cz) * g.cell, zhi = g.cell - zlo;
    float bound = limit2 * 1.0001f + 1e-12f;
    if (x0 <= x1) {
#pragma unroll 1
      for (int it = sub; it < 9; it += kGroup) {
        const int pass = it == 0 ? 4 : (it == 4 ? 0 : it);              // home row (dy = dz = 0) first
        const int dy = pass % 3 - 1, dz = pass / 3 - 1;
        const float ey = dy < 0 ? ylo : (dy > 0 ? yhi : 0.f), ez = dz < 0 ? zlo : (dz > 0 ? zhi : 0.f);
        if (ey * ey + ez * ez > bound) continue;
        const int y = iy + dy, z = iz + dz;
        if (y < 0 || y >= g.dim[1] || z < 0 || z >= g.dim[2]) continue;
        const int row = (z * g.dim[1] + y) * g.dim[0];
        const int s0 = g.cell_start[row + x0], s1 = g.cell_start[row + x1 + 1];
        for (int s = s0; s < s1; s++) {
          const float4 p = g.pts[s];
          const float dx = qx - p.x, dy2 = qy - p.y, dz2 = qz - p.z;
          const float d = ((dx * dx) + dy2 * dy2) + dz2 * dz2;
          const int idx = __float_as_int(p.w);
          if (d < bd || (d == bd && idx < best)) {
            bd = d;
            best = idx;
          }
        }
        bound = fminf(bound, bd * 1.0001f + 1e-12f);                     // later rows must beat the best so far
      }
    }
  }
  // lexicographic (distance, index) minimum over the group; -1 (nothing found) loses against any hit
#pragma unroll
  for (int off = 1; off < kGroup; off <<= 1) {
    const float od = __shfl_xor(bd, off, kGroup);
    const int oi = __shfl_xor(best, off, kGroup);
    if (oi >= 0 && (best < 0 || od < bd || (od == bd && oi < best))) {
      bd = od;
      best = oi;
    }
  }
  best_d = bd;
  return best;
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

// Registration pre-check, CorresApp.cpp:257-264.  kGroup lanes per source point; a fixed grid strides over the
// points and issues ONE atomic per workgroup (one per wave serialised thousands of atomics on one word).
__global__ __launch_bounds__(kBlock) void k_count_inliers(const float4* __restrict__ src_sorted, int n, Mat12d T, Grid g, float radius,
                                                          double maxd2, int* __restrict__ count) {
  const int sub = threadIdx.x % kGroup;
  int local = 0;
  for (int k = (blockIdx.x * blockDim.x + threadIdx.x) / kGroup; k < n; k += gridDim.x * (blockDim.x / kGroup)) {
    float qx, qy, qz, d;
    const float4 s = src_sorted[k];
    xform_d(T, s.x, s.y, s.z, qx, qy, qz);
    const int i = nn_search(g, qx, qy, qz, radius * radius, sub, d);
    if (sub == 0 && i >= 0 && (double)d <= (double)radius * (double)radius && (double)d < maxd2) local++;
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

// One ICP iteration, part 1 (kGroup lanes per point): X <- delta * X (the previous iteration's increment,
// float32) and correspondence estimation: nn[k] = target index kept if d^2 <= max_dist^2, else -1.
__global__ __launch_bounds__(kBlock) void k_icp_nn(float* __restrict__ X, int n, Mat12f delta, int apply, Grid g, float radius,
                                                   double maxd2, int* __restrict__ nn, float* __restrict__ nd) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = t / kGroup, sub = t % kGroup;
  if (k >= n) return;
  float sx = X[3 * k], sy = X[3 * k + 1], sz = X[3 * k + 2];
  if (apply) {
    const float x = sx, y = sy, z = sz;
    sx = ((delta.m[0] * x + delta.m[1] * y) + delta.m[2] * z) + delta.m[3];
    sy = ((delta.m[4] * x + delta.m[5] * y) + delta.m[6] * z) + delta.m[7];
    sz = ((delta.m[8] * x + delta.m[9] * y) + delta.m[10] * z) + delta.m[11];
  }
  float d;
  const int i = nn_search(g, sx, sy, sz, radius * radius, sub, d);   // (all lanes read X before anybody overwrites it)
  if (sub == 0) {
    if (apply) {
      X[3 * k] = sx;
      X[3 * k + 1] = sy;
      X[3 * k + 2] = sz;
    }
    const bool keep = i >= 0 && (double)d <= (double)radius * (double)radius && !((double)d > maxd2);
    nn[k] = keep ? i : -1;
    nd[k] = d;
  }
}

// One ICP iteration, part 2 (one lane per point): the sums of TransformationEstimationPointToPlaneLLS.
// acc: [0..20] upper triangle of AtA row by row, [21..26] Atb, [27] sum of d^2, [28] count.
__global__ __launch_bounds__(kBlock) void k_icp_accum(const float* __restrict__ X, int n, const int* __restrict__ nn,
                                                      const float* __restrict__ nd, const float* __restrict__ tgt_xyz,
                                                      const float* __restrict__ tgt_nrm, double* __restrict__ acc) {
  double v[29];
#pragma unroll
  for (int i = 0; i < 29; i++) v[i] = 0.0;
  // a fixed grid strides over the points: each thread folds several rows before the (expensive) 29-value reduction
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const int i = nn[k];
    if (i >= 0) {
      const float sx = X[3 * k], sy = X[3 * k + 1], sz = X[3 * k + 2];
      const float dx = tgt_xyz[3 * i], dy = tgt_xyz[3 * i + 1], dz = tgt_xyz[3 * i + 2];
      const float nx = tgt_nrm[3 * i], ny = tgt_nrm[3 * i + 1], nz = tgt_nrm[3 * i + 2];
      v[27] += (double)nd[k];
      v[28] += 1.0;
      if (isfinite(sx) && isfinite(sy) && isfinite(sz) && isfinite(nx) && isfinite(ny) && isfinite(nz)) {
        const double a = (double)(nz * sy - ny * sz);      // float32 expressions widened to double (PCL)
        const double b = (double)(nx * sz - nz * sx);
        const double c = (double)(ny * sx - nx * sy);
        const double dnx = nx, dny = ny, dnz = nz;
        v[0] += a * a;  v[1] += a * b;  v[2] += a * c;  v[3] += a * dnx;  v[4] += a * dny;  v[5] += a * dnz;
        v[6] += b * b;  v[7] += b * c;  v[8] += b * dnx; v[9] += b * dny; v[10] += b * dnz;
        v[11] += c * c; v[12] += c * dnx; v[13] += c * dny; v[14] += c * dnz;
        v[15] += dnx * dnx; v[16] += dnx * dny; v[17] += dnx * dnz;
        v[18] += dny * dny; v[19] += dny * dnz;
        v[20] += dnz * dnz;
        const double e = (double)(nx * dx + ny * dy + nz * dz - nx * sx - ny * sy - nz * sz);
        v[21] += a * e; v[22] += b * e; v[23] += c * e; v[24] += dnx * e; v[25] += dny * e; v[26] += dnz * e;
      }
    }
  }
  block_reduce_atomic<29>(v, acc);
}

// getFitnessScore-style diagnostic: squared NN distance of final * source inside the search radius (-1 = none).
__global__ __launch_bounds__(kBlock) void k_fitness_nn(const float4* __restrict__ src_sorted, int n, Mat12f M, Grid g, float radius,
                                                       float* __restrict__ nd) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int k = t / kGroup, sub = t % kGroup;
  if (k >= n) return;
  const float4 s = src_sorted[k];
  const float x = s.x, y = s.y, z = s.z;
  const float qx = ((M.m[0] * x + M.m[1] * y) + M.m[2] * z) + M.m[3];
  const float qy = ((M.m[4] * x + M.m[5] * y) + M.m[6] * z) + M.m[7];
  const float qz = ((M.m[8] * x + M.m[9] * y) + M.m[10] * z) + M.m[11];
  float d;
  const int i = nn_search(g, qx, qy, qz, radius * radius, sub, d);
  if (sub == 0) nd[k] = (i >= 0 && (double)d <= (double)radius * (double)radius) ? d : -1.0f;
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

// FindCorrespondence, CorresApp.cpp:144-161 (kGroup lanes per source point): match[original index] = NN index
// passing the distance and normal tests, else -1.
__global__ __launch_bounds__(kBlock) void k_find_corr(const float4* __restrict__ src_sorted, const float* __restrict__ nrm, int n,
                                                      Mat12d T, Grid g, const float* __restrict__ tgt_nrm, float radius,
                                                      double dist2, double normal_cos, int* __restrict__ match) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int q = t / kGroup, sub = t % kGroup;                  // q = position in the source's cell-sorted order
  if (q >= n) return;
  const float4 s = src_sorted[q];
  const int k = __float_as_int(s.w);                           // original (file-order) index of this source point
  float qx, qy, qz, d;
  xform_d(T, s.x, s.y, s.z, qx, qy, qz);
  const int i = nn_search(g, qx, qy, qz, radius * radius, sub, d);
  if (sub != 0) return;
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

// ---- host-side small algebra -------------------------------------------------------------------
// Dense 6x6 solve by Gaussian elimination with partial pivoting (PCL: ATA.inverse() * ATb).
bool solve6x6(double A[6][6], double b[6], double x[6]) {
  for (int c = 0; c < 6; c++) {
    int p = c;
    for (int r = c + 1; r < 6; r++)
      if (std::fabs(A[r][c]) > std::fabs(A[p][c])) p = r;
    if (A[p][c] == 0.0 || !std::isfinite(A[p][c])) return false;
    if (p != c) {
      for (int k = 0; k < 6; k++) std::swap(A[p][k], A[c][k]);
      std::swap(b[p], b[c]);
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

// TransformationEstimationPointToPlaneLLS::constructTransformationMatrix: Rz(gamma) Ry(beta) Rx(alpha), float storage.
void construct_increment(const double x[6], float M[16]) {
  const double al = x[0], be = x[1], ga = x[2];
  for (int i = 0; i < 16; i++) M[i] = 0.f;
  M[0] = (float)(cos(ga) * cos(be));
  M[1] = (float)(-sin(ga) * cos(al) + cos(ga) * sin(be) * sin(al));
  M[2] = (float)(sin(ga) * sin(al) + cos(ga) * sin(be) * cos(al));
  M[4] = (float)(sin(ga) * cos(be));
  M[5] = (float)(cos(ga) * cos(al) + sin(ga) * sin(be) * sin(al));
  M[6] = (float)(-cos(ga) * sin(al) + sin(ga) * sin(be) * cos(al));
  M[8] = (float)(-sin(be));
  M[9] = (float)(cos(be) * sin(al));
  M[10] = (float)(cos(be) * cos(al));
  M[3] = (float)x[3];
  M[7] = (float)x[4];
  M[11] = (float)x[5];
  M[15] = 1.f;
}

void mul4f(const float* A, const float* B, float* C) {
  float t[16];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++)
      t[r * 4 + c] = ((A[r * 4] * B[c] + A[r * 4 + 1] * B[4 + c]) + A[r * 4 + 2] * B[8 + c]) + A[r * 4 + 3] * B[12 + c];
  memcpy(C, t, sizeof t);
}

}  // namespace

struct er_cloud_s {
  int device = 0, n = 0;
  float *xyz = nullptr, *nrm = nullptr;
  float4* sorted = nullptr;
  int* cell_start = nullptr;
  Grid grid{};
  float radius_cap = 0.f;       // largest search radius the grid supports
  hipStream_t stream = nullptr;
  // scratch for the SOURCE role
  std::mutex src_mutex;
  float *X = nullptr, *nd = nullptr;
  int *match = nullptr, *block_count = nullptr, *block_offset = nullptr, *pairs = nullptr, *icount = nullptr;
  double* acc = nullptr;
  int nblocks = 0;              // one lane per point
  int gblocks = 0;              // kGroup lanes per point
};

namespace {
Grid grid_of(const er_cloud_s* c) { return c->grid; }
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
  // ---- uniform grid on the host (once per fragment; every pair that uses it as target reuses it) ----
  float cell = grid_cell * 1.001f;               // strictly larger than any admissible radius
  float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  if (n > 0) {
    for (int a = 0; a < 3; a++) lo[a] = FLT_MAX, hi[a] = -FLT_MAX;
    for (int i = 0; i < n; i++)
      for (int a = 0; a < 3; a++) {
        const float v = xyz_host[3 * (size_t)i + a];
        if (v < lo[a]) lo[a] = v;
        if (v > hi[a]) hi[a] = v;
      }
    for (int a = 0; a < 3; a++)
      if (!std::isfinite(lo[a]) || !std::isfinite(hi[a])) {
        delete c;
        return er::fail("er_cloud_create: non-finite coordinates");
      }
  }
  int dim[3];
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
  std::vector<int> cs((size_t)ncell + 1, 0), id((size_t)n);
  for (int i = 0; i < n; i++) {
    int q[3];
    for (int a = 0; a < 3; a++) {
      q[a] = (int)std::floor((xyz_host[3 * (size_t)i + a] - lo[a]) / cell);
      q[a] = std::min(std::max(q[a], 0), dim[a] - 1);
    }
    id[(size_t)i] = (q[2] * dim[1] + q[1]) * dim[0] + q[0];
    cs[(size_t)id[(size_t)i] + 1]++;
  }
  for (int k = 0; k < ncell; k++) cs[(size_t)k + 1] += cs[(size_t)k];
  std::vector<float4> sorted((size_t)n);
  {
    std::vector<int> fill(cs.begin(), cs.end() - 1);
    for (int i = 0; i < n; i++) {
      const int s = fill[(size_t)id[(size_t)i]]++;
      float w;
      memcpy(&w, &i, sizeof w);
      sorted[(size_t)s] = make_float4(xyz_host[3 * (size_t)i], xyz_host[3 * (size_t)i + 1], xyz_host[3 * (size_t)i + 2], w);
    }
  }
  const size_t nn = (size_t)std::max(n, 1);
  c->nblocks = (int)((nn + kBlock - 1) / kBlock);
  c->gblocks = (int)((nn * kGroup + kBlock - 1) / kBlock);
#define ER_CALLOC(ptr, bytes)                                                                 \
  do {                                                                                        \
    hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                                       \
    if (e_ != hipSuccess) {                                                                   \
      er::fail("er_cloud_create: hipMalloc(%zu) failed: %s", (size_t)(bytes), hipGetErrorString(e_)); \
      er_cloud_destroy(c);                                                                    \
      return 1;                                                                               \
    }                                                                                         \
  } while (0)
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return er::fail("er_cloud_create: hipStreamCreate failed");
  }
  ER_CALLOC(c->xyz, nn * 3 * sizeof(float));
  ER_CALLOC(c->nrm, nn * 3 * sizeof(float));
  ER_CALLOC(c->sorted, nn * sizeof(float4));
  ER_CALLOC(c->cell_start, ((size_t)ncell + 1) * sizeof(int));
  ER_CALLOC(c->X, nn * 3 * sizeof(float));
  ER_CALLOC(c->match, nn * sizeof(int));
  ER_CALLOC(c->nd, nn * sizeof(float));
  ER_CALLOC(c->pairs, nn * 2 * sizeof(int));
  ER_CALLOC(c->block_count, (size_t)c->nblocks * sizeof(int));
  ER_CALLOC(c->block_offset, (size_t)c->nblocks * sizeof(int));
  ER_CALLOC(c->icount, 4 * sizeof(int));
  ER_CALLOC(c->acc, kAcc * sizeof(double));
#undef ER_CALLOC
  bool ok = true;
  if (n > 0) {
    ok = ok && hipMemcpy(c->xyz, xyz_host, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(c->nrm, normal_host, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(c->sorted, sorted.data(), (size_t)n * sizeof(float4), hipMemcpyHostToDevice) == hipSuccess;
  }
  ok = ok && hipMemcpy(c->cell_start, cs.data(), cs.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) {
    er_cloud_destroy(c);
    return er::fail("er_cloud_create: upload failed: %s", hipGetErrorString(hipGetLastError()));
  }
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
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  void* ptrs[] = {c->xyz, c->nrm, c->sorted, c->cell_start, c->X, c->nd, c->match, c->pairs, c->block_count, c->block_offset, c->icount, c->acc};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

int er_cloud_size(er_cloud_t c) { return c ? c->n : -1; }

static int check_pair(er_cloud_t src, er_cloud_t tgt, double radius, const char* who) {
  if (!src || !tgt) return er::fail("%s: NULL cloud", who);
  if (src->device != tgt->device) return er::fail("%s: source and target live on different devices", who);
  if (!(radius > 0.0) || radius > (double)tgt->radius_cap * (1.0 + 1e-6))
    return er::fail("%s: search radius %g exceeds the target's grid cell %g (er_cloud_create grid_cell)", who, radius, (double)tgt->radius_cap);
  return 0;
}

int er_icp_count_inliers(er_cloud_t src, er_cloud_t tgt, const double T[16], double max_dist, int* count) {
  if (!T || !count) return er::fail("er_icp_count_inliers: NULL argument");
  if (check_pair(src, tgt, max_dist, "er_icp_count_inliers")) return 1;
  ER_HIP_TRY(hipSetDevice(src->device));
  std::lock_guard<std::mutex> lock(src->src_mutex);
  Mat12d M;
  for (int q = 0; q < 12; q++) M.m[q] = T[q];
  ER_HIP_TRY(hipMemsetAsync(src->icount, 0, sizeof(int), src->stream));
  if (src->n > 0 && tgt->n > 0) {
    hipLaunchKernelGGL(k_count_inliers, dim3(std::min(src->gblocks, 2048)), dim3(kBlock), 0, src->stream, src->sorted, src->n, M, grid_of(tgt),
                       (float)max_dist, max_dist * max_dist, src->icount);
    ER_HIP_TRY(hipGetLastError());
  }
  ER_HIP_TRY(hipMemcpyAsync(count, src->icount, sizeof(int), hipMemcpyDeviceToHost, src->stream));
  ER_HIP_TRY(hipStreamSynchronize(src->stream));
  return 0;
}

int er_icp_align(er_cloud_t src, er_cloud_t tgt, const float guess[16], double max_dist, int max_iter,
                 double transformation_epsilon, int stop_rule, float out[16], int* iterations, int* converged,
                 double* fitness) {
  if (!guess || !out) return er::fail("er_icp_align: NULL argument");
  if (check_pair(src, tgt, max_dist, "er_icp_align")) return 1;
  ER_HIP_TRY(hipSetDevice(src->device));
  std::lock_guard<std::mutex> lock(src->src_mutex);
  const int n = src->n;
  float fin[16];
  memcpy(fin, guess, sizeof fin);                          // final_transformation_ = guess
  bool ident = true;
  for (int i = 0; i < 16; i++) ident = ident && guess[i] == ((i % 5 == 0) ? 1.f : 0.f);
  Mat12f G;
  for (int q = 0; q < 12; q++) G.m[q] = guess[q];
  if (n > 0) {
    hipLaunchKernelGGL(k_init_x, dim3(src->nblocks), dim3(kBlock), 0, src->stream, src->sorted, src->X, n, G, ident ? 0 : 1);
    ER_HIP_TRY(hipGetLastError());
  }
  float delta[16], prev_delta[16];
  for (int i = 0; i < 16; i++) delta[i] = prev_delta[i] = (i % 5 == 0) ? 1.f : 0.f;
  int iter = 0;
  bool conv = false;
  double prev_mse = DBL_MAX;
  const double maxd2 = max_dist * max_dist;
  const Grid g = grid_of(tgt);
  for (;;) {
    memcpy(prev_delta, delta, sizeof delta);
    Mat12f D;
    for (int q = 0; q < 12; q++) D.m[q] = delta[q];
    double acc[kAcc];
    ER_HIP_TRY(hipMemsetAsync(src->acc, 0, kAcc * sizeof(double), src->stream));
    if (n > 0 && tgt->n > 0) {
      hipLaunchKernelGGL(k_icp_nn, dim3(src->gblocks), dim3(kBlock), 0, src->stream, src->X, n, D, iter > 0 ? 1 : 0, g,
                         (float)max_dist, maxd2, src->match, src->nd);
      hipLaunchKernelGGL(k_icp_accum, dim3(std::min(src->nblocks, 256)), dim3(kBlock), 0, src->stream, src->X, n, src->match, src->nd, tgt->xyz,
                         tgt->nrm, src->acc);
      ER_HIP_TRY(hipGetLastError());
    }
    ER_HIP_TRY(hipMemcpyAsync(acc, src->acc, kAcc * sizeof(double), hipMemcpyDeviceToHost, src->stream));
    ER_HIP_TRY(hipStreamSynchronize(src->stream));
    const double cnt = acc[28];
    if (cnt < 3.0) { conv = false; break; }                // min_number_correspondences_
    double A[6][6], b[6], x[6];
    int t = 0;
    for (int r = 0; r < 6; r++)
      for (int c2 = r; c2 < 6; c2++) A[r][c2] = A[c2][r] = acc[t++];
    for (int r = 0; r < 6; r++) b[r] = acc[21 + r];
    if (!solve6x6(A, b, x)) { conv = false; break; }
    construct_increment(x, delta);
    mul4f(delta, fin, fin);                                // final = increment * final
    ++iter;
    if (iter >= max_iter) { conv = true; break; }
    if (stop_rule == 0) {                                  // PCL 1.7 DefaultConvergenceCriteria
      const double cos_angle = 0.5 * (double)(delta[0] + delta[5] + delta[10] - 1.f);
      const double tr2 = (double)(delta[3] * delta[3] + delta[7] * delta[7] + delta[11] * delta[11]);
      if (cos_angle >= 1.0 - transformation_epsilon && tr2 <= transformation_epsilon) { conv = true; break; }
      const double cur = acc[27] / cnt;
      if (std::fabs(cur - prev_mse) < 1e-12) { conv = true; break; }
      prev_mse = cur;
    } else {                                               // PCL <= 1.6
      float s = 0.f;
      for (int i = 0; i < 16; i++) s += delta[i] - prev_delta[i];
      if (std::fabs((double)s) < transformation_epsilon) { conv = true; break; }
    }
  }
  memcpy(out, fin, sizeof fin);
  if (iterations) *iterations = iter;
  if (converged) *converged = conv ? 1 : 0;
  if (fitness) {
    Mat12f F;
    for (int q = 0; q < 12; q++) F.m[q] = fin[q];
    double acc[2] = {0, 0};
    ER_HIP_TRY(hipMemsetAsync(src->acc, 0, kAcc * sizeof(double), src->stream));
    if (n > 0 && tgt->n > 0) {
      hipLaunchKernelGGL(k_fitness_nn, dim3(src->gblocks), dim3(kBlock), 0, src->stream, src->sorted, n, F, g, (float)max_dist, src->nd);
      hipLaunchKernelGGL(k_fitness_sum, dim3(src->nblocks), dim3(kBlock), 0, src->stream, src->nd, n, src->acc);
      ER_HIP_TRY(hipGetLastError());
    }
    ER_HIP_TRY(hipMemcpyAsync(acc, src->acc, 2 * sizeof(double), hipMemcpyDeviceToHost, src->stream));
    ER_HIP_TRY(hipStreamSynchronize(src->stream));
    *fitness = acc[1] > 0 ? acc[0] / acc[1] : DBL_MAX;
  }
  return 0;
}

int er_find_correspondence(er_cloud_t src, er_cloud_t tgt, const double T[16], double dist, double normal_cos,
                           int* pairs_host, int capacity, int* n_pairs, double* info36) {
  if (!T || !n_pairs || (capacity > 0 && !pairs_host)) return er::fail("er_find_correspondence: NULL argument");
  if (check_pair(src, tgt, dist, "er_find_correspondence")) return 1;
  ER_HIP_TRY(hipSetDevice(src->device));
  std::lock_guard<std::mutex> lock(src->src_mutex);
  const int n = src->n;
  Mat12d M;
  for (int q = 0; q < 12; q++) M.m[q] = T[q];
  *n_pairs = 0;
  if (info36) memset(info36, 0, 36 * sizeof(double));
  if (n == 0 || tgt->n == 0) return 0;
  ER_HIP_TRY(hipMemsetAsync(src->acc, 0, kAcc * sizeof(double), src->stream));
  hipLaunchKernelGGL(k_find_corr, dim3(src->gblocks), dim3(kBlock), 0, src->stream, src->sorted, src->nrm, n, M, grid_of(tgt),
                     tgt->nrm, (float)dist, dist * dist, normal_cos, src->match);
  hipLaunchKernelGGL(k_count_blocks, dim3(src->nblocks), dim3(kBlock), 0, src->stream, src->match, src->xyz, n, src->block_count,
                     src->acc, info36 ? 1 : 0);
  hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, src->stream, src->block_count, src->block_offset, src->nblocks, src->icount + 1);
  hipLaunchKernelGGL(k_compact, dim3(src->nblocks), dim3(kBlock), 0, src->stream, src->match, n, src->block_offset, src->pairs, n);
  ER_HIP_TRY(hipGetLastError());
  int total = 0;
  double acc[10];
  ER_HIP_TRY(hipMemcpyAsync(&total, src->icount + 1, sizeof(int), hipMemcpyDeviceToHost, src->stream));
  ER_HIP_TRY(hipMemcpyAsync(acc, src->acc, sizeof acc, hipMemcpyDeviceToHost, src->stream));
  ER_HIP_TRY(hipStreamSynchronize(src->stream));
  *n_pairs = total;
  const int ncopy = std::min(total, capacity);
  if (ncopy > 0) {
    ER_HIP_TRY(hipMemcpyAsync(pairs_host, src->pairs, (size_t)ncopy * 2 * sizeof(int), hipMemcpyDeviceToHost, src->stream));
    ER_HIP_TRY(hipStreamSynchronize(src->stream));
  }
  if (info36) {
    // sum A^T A with A = [I | B], B = [[0, 2sz, -2sy], [-2sz, 0, 2sx], [2sy, -2sx, 0]]  (CorresApp.cpp:198-203)
    double* I = info36;
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
  return total > capacity ? er::fail("er_find_correspondence: %d pairs exceed the capacity %d", total, capacity) : 0;
}

}  // extern "C"
