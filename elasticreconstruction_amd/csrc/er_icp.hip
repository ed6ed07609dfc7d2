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
// plus GROUP workspaces (streams, per-pair descriptors and ICP states, scratch slabs cut into one slice per pair) borrowed from a
// per-device pool, so clouds are immutable and any number of pair lists can be in flight.
// Queries run in the SOURCE cloud's own cell-sorted order (thread t takes sorted[t]), so the lanes of a wave
// walk the same few target cells together (coalesced / broadcast candidate loads); results are written back
// by original index.  The NN search is block-cooperative (see nn_block); candidates stream from L2 as 16-byte loads.
// Every stage processes a whole GROUP of pairs per launch (blockIdx.y = pair; "pair groups" below):
//   k_count_inliers   transform (float64 -> float32) + NN + count           (Registration pre-check)
//   k_icp_iter        one ICP iteration: [apply guess / last increment] + NN + point-to-plane rows -> 27+2 float64 sums
//                     (4 points per thread -> wave shuffle -> LDS -> per-workgroup partial)
//   k_icp_final       one workgroup per pair: fixed-order total, 6x6 solve, increment, PCL's stop rule -- the loop stays on the device
//   k_find_corr       transform points+normals + NN + distance/normal tests -> match[orig index];
//                     k_count_blocks (+ information-matrix sums) + k_scan_blocks + k_compact = stable compaction in file order
// Reductions and scans, not contractions: no MFMA.
#include "er_common.h"

#include "../../include/er_hip.h"

#include <hipcub/hipcub.hpp>   // device radix sort / prefix sum of the grid build (library primitives; everything else is hand-written)

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kBlock = 256;
constexpr int kAcc = 32;   // 21 ATA + 6 ATb + sum d^2 + count (+ padding)

// PairDev carries its pointers through memory, so the compiler emits FLAT loads for them (64-bit VALU address arithmetic, both wait counters).
// The hot accesses go through explicitly GLOBAL pointer types: global_load with a scalar base and a 32-bit offset, vmcnt only.
#ifndef ER_ICP_GLOBAL_PTR
#define ER_ICP_GLOBAL_PTR 1
#endif
typedef float f4v __attribute__((ext_vector_type(4)));
#if ER_ICP_GLOBAL_PTR
#define ER_GLOBAL __attribute__((address_space(1)))
#else
#define ER_GLOBAL
#endif
typedef const ER_GLOBAL f4v* gp_f4;
typedef const ER_GLOBAL float* gp_f;
typedef const ER_GLOBAL int* gp_i;
typedef ER_GLOBAL float* gp_fw;
typedef ER_GLOBAL int* gp_iw;
#define ER_GP(type, ptr) ((type)(ptr))

// The grid carries TWO RINGS OF EMPTY CELLS around the cloud's bounding box (cell (x, y, z) of the box is cell (x + 2, y + 2, z + 2) of the array).  A query is
// searched only if its home cell lies within one cell of the box, so every cell of its 27-neighbourhood EXISTS -- no flags, range tests or clamps -- and
// the four bounds L, O, R, E of a row's three cells x-1, x, x+1 are four consecutive ints of cell_start: one 16-byte load per row.
// (What was tried on this search and dropped, with numbers: profiles/HISTORY.md "Path B: the search, rounds 3-6".)
#ifndef ER_NN_PAD_BATCH
#define ER_NN_PAD_BATCH 4      // rows whose four bounds are in flight at a time (4 registers per row; 8 = all rows: +11 VGPRs, within the noise)
#endif
typedef int i4v __attribute__((ext_vector_type(4), aligned(4)));   // (the four bounds of a row start at an arbitrary cell: 4-byte alignment only)
struct Grid {
  const float4* pts;
  const int* cell_start;
  float org[3];
  float cell;
  int dim[3];      // cells of the bounding box per axis (the array has dim + 4 per axis: two rings of empty cells)
  float slack;     // absolute part of nn_block's pruning margin (square metres), from the grid's extent: grid_slack()
  int pnx, pny;    // dim[0] + 4, dim[1] + 4: strides of the padded array
  const unsigned char* occ;   // [cells] 1 = some cell of this cell's 27-neighbourhood holds a point (round 6; see ER_NN_OCC)
};

// The pruning margin of nn_block.  A cell (or row of cells) is skipped when the squared distance f'^2 from the query to its nearest face, as the
// kernel computes it, exceeds  B * (1 + 1e-4) + slack,  B = the best float32 squared distance so far (or the squared search radius).  For that to
// be exact -- no point p of a skipped cell may have a float32 distance below B -- the margin has to cover what the float32 cell arithmetic can be
// off by:  u = fl(fl(q - org) / cell) carries a relative error of 2 x 2^-24, i.e. up to 1.2e-7 x |q - org| metres in the face distance, and the
// target points were assigned to their cells by the same expression, so the face itself is that fuzzy once more:  f' <= f + D  per axis with
// D = 2.5e-7 x (largest extent + 2 cells) + 4e-9, and sqrt(3) D for the rows and corners that combine two or three axes.  With S^2 = B (1 + r) + A,
// a skipped point has a true distance >= S - sqrt(3) D, its float32 squared distance is >= (S - sqrt(3) D)^2 (1 - 3e-7), and
// 2 S sqrt(3) D <= (r / 4) S^2 + 12 D^2 / r  gives  (S - sqrt(3) D)^2 (1 - 3e-7) >= B  as soon as  A >= 1.2001e5 D^2  (r = 1e-4); the code takes 1.3e5.
// (A constant absolute part would cover D only for best distances below a micrometre or above several millimetres; tests/test_icp_gpu.py builds the
// queries in between on purpose.)
inline float grid_slack(const int dim[3], float cell) {
  const int big = std::max(dim[0], std::max(dim[1], dim[2]));
  const double D = 2.5e-7 * (double)(big + 2) * (double)cell + 4e-9;
  return (float)(1.3e5 * D * D);
}

struct Mat12d { double m[12]; };
struct Mat12f { float m[12]; };

// Exact 1-NN of q among the target points of the 27 neighbouring cells: float32 squared distance ((dx*dx) + dy*dy) + dz*dz (FLANN L2_Simple), ties ->
// lower original index.  limit2 = squared search radius: callers discard anything farther, so cells lying entirely beyond it are skipped (margin: 1e-4
// relative plus an absolute part sized from the grid's extent, see grid_slack).
// Block-cooperative, two phases (one query per thread, kBlock queries per workgroup):
//   phase 0  the thread scans its query's own cell, then the left / right cell of the home row if its face is closer than the best so far; the eight
//            neighbour rows are trimmed by the same face tests and their NON-EMPTY ranges go to an LDS task list as (first candidate, count, query);
//   phase 1  the workgroup shares the list -- one range per thread and trip -- folding results into the query's packed (distance bits, index) key with a
//            64-bit LDS atomicMin, which IS the lexicographic (distance, index) minimum.
// Why: ~1.7 of the 8 neighbour rows survive for an average query, but a SIMT loop runs every row ANY lane needs; compacting the survivors across the
// workgroup removes that waste.  Candidates are scanned kUnroll at a time (the 16-byte loads are issued together).  What bounds it (counters,
// profiles/r04w_*, r05f_*): VALU issue in the straight-line part every query runs (~360 of ~735 instructions per slice of 64 queries) and LDS traffic per task.
#ifndef ER_ICP_UNROLL
#define ER_ICP_UNROLL 4
#endif
constexpr int kUnroll = ER_ICP_UNROLL;
constexpr unsigned long long kNoHit = ((unsigned long long)0x7f7fffffu << 32) | 0xffffffffull;   // (FLT_MAX, -1)

#ifndef ER_NN_TASKCAP
#define ER_NN_TASKCAP (kBlock * 4)
#endif
#ifndef ER_NN_OCC
#define ER_NN_OCC 1
#endif
constexpr int kTaskCap = ER_NN_TASKCAP;   // (query, row) tasks of phase 1 held in LDS; a task beyond that is scanned by the thread that found it

struct NnShared {
  unsigned long long best[kBlock];
  float q[3][kBlock];
  int task_s0[kTaskCap];          // first candidate of the task's range in the cell-sorted target
  int task_nq[kTaskCap];          // candidates << 8 | query
  int ntask;
};

// Candidates [s0, s1) of the cell-sorted target against the query: the packed (distance bits, index) minimum -- every candidate against the key, exact by
// construction; ~54 VALU instructions per trip of four, 20 of them the selection.
// The kU loads of a trip are NOT clamped to the range: a trip that starts inside [s0, s1) may read up to kU - 1 entries past s1.  Those are points of the
// cells that follow in the cell-sorted array -- REAL points of the same cloud, so the minimum over the wider set is still the exact nearest neighbour --
// or, behind the cloud's last point, the kSentinel entries of +inf er_cloud_create appends (distance inf / NaN: a bit pattern that never wins).
#ifndef ER_NN_NOCLAMP
#define ER_NN_NOCLAMP 1
#endif
constexpr int kSentinel = 8;            // float4 entries of +inf behind every cloud's sorted array (>= the largest kU - 1)
template <int kU = kUnroll>
__device__ __forceinline__ unsigned long long scan_range(const Grid& g, int s0, int s1, float qx, float qy, float qz, unsigned long long key) {
  // candidates are addressed by UNSIGNED 32-bit byte offsets from the (wave-uniform) base: a scalar-base global load and one 32-bit
  // add per candidate instead of a sign extension and a 64-bit multiply-add each
  const ER_GLOBAL char* base = (const ER_GLOBAL char*)g.pts;
  const unsigned last = (unsigned)(s1 - 1) * 16u;
  for (unsigned o = (unsigned)s0 * 16u; o <= last && s0 < s1; o += 16u * kU) {
    f4v p[kU];
#pragma unroll
#if ER_NN_NOCLAMP
    for (int u = 0; u < kU; u++) p[u] = *(const ER_GLOBAL f4v*)(base + o + 16u * (unsigned)u);
#else
    for (int u = 0; u < kU; u++) p[u] = *(const ER_GLOBAL f4v*)(base + min(o + 16u * (unsigned)u, last));
#endif
#pragma unroll
    for (int u = 0; u < kU; u++) {
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
// Returns the index (or -1) and the squared distance of the nearest target point.  A cell is skipped only when every point in it is provably
// farther than the best so far: its nearest face already is, with the 1e-4 relative margin of grid_slack; ties cannot hide there.
// hit2 >= 0 (the Registration pre-check): the caller only needs to know whether ANY target point lies closer than sqrt(hit2) -- "count the points
// whose nearest neighbour is within reg_dist" is "count the points that have a neighbour within reg_dist" -- so a query stops as soon as it holds a
// candidate below hit2; the returned distance is then that candidate's, not the minimum.  Queries without one run the full exact search.
// kU = candidates per trip: 4 where the nearest neighbour is needed, 3 in the any-hit pre-check (profiles/r04t_ab_scan_unroll.txt).
template <int kU = kUnroll>
__device__ __forceinline__ int nn_block(NnShared& sh, const Grid& g, bool active, float qx, float qy, float qz, float limit2,
                                        float& best_d, float hit2 = -1.f) {
  const int tid = threadIdx.x;
  __syncthreads();                                            // the previous call's readers are done with `sh`
  if (tid == 0) sh.ntask = 0;
  // the query's cell (float32 expressions shared with the grid build)
  const float ux = (qx - g.org[0]) / g.cell, uy = (qy - g.org[1]) / g.cell, uz = (qz - g.org[2]) / g.cell;
  const float cx = floorf(ux), cy = floorf(uy), cz = floorf(uz);
  const bool inside = active && cx >= -1.f && cx <= (float)g.dim[0] && cy >= -1.f && cy <= (float)g.dim[1] && cz >= -1.f && cz <= (float)g.dim[2];
  const int ix = inside ? (int)cx : 0, iy = inside ? (int)cy : 0, iz = inside ? (int)cz : 0;
  __syncthreads();                                            // (sh.ntask is zero)
  unsigned long long key = kNoHit;
#if ER_NN_OCC
  // A query whose whole 27-neighbourhood is empty (55-63 % of the queries of a fragment pair: the part of the source that does not overlap the target)
  // has no neighbour within the radius; one byte per cell (k_chunk_occ) says so before the row tests.  Queries come in the source's cell order: whole
  // waves leave here (9-10 % of the ICP phase, profiles/r06c_*).
  const bool live = inside && ER_GP(const ER_GLOBAL unsigned char*, g.occ)[(unsigned)(((iz + 2) * g.pny + (iy + 2)) * g.pnx + (ix + 2))] != 0;
#else
  const bool live = inside;
#endif
  if (live) {
    // distance from q to the lower / upper face of its own cell along x, y and z (metres), squared
    const float xlo = (ux - cx) * g.cell, xhi = g.cell - xlo, ylo = (uy - cy) * g.cell, yhi = g.cell - ylo, zlo = (uz - cz) * g.cell,
                zhi = g.cell - zlo;
    const float xl2 = xlo * xlo, xr2 = xhi * xhi;
    float bound = limit2 * 1.0001f + g.slack;
    const ER_GLOBAL char* csb = (const ER_GLOBAL char*)g.cell_start;
    const int pnx = g.pnx, pny = g.pny;
    int home = ((iz + 2) * pny + (iy + 2)) * pnx + (ix + 1);   // the cell LEFT of the query's own cell: where the four bounds of a row's three cells begin
    {
      const i4v h = *(const ER_GLOBAL i4v*)(csb + (unsigned)home * 4u);   // L, O, R, E of the home row
      key = scan_range<kU>(g, h.y, h.z, qx, qy, qz, key);                  // the query's own cell first
      bound = fminf(bound, __uint_as_float((unsigned)(key >> 32)) * 1.0001f + g.slack);   // the other cells must beat this one
      if (__uint_as_float((unsigned)(key >> 32)) < hit2) bound = -1.f;                   // (any-hit mode: done)
      if (xl2 <= bound) {
        key = scan_range<kU>(g, h.x, h.y, qx, qy, qz, key);
        bound = fminf(bound, __uint_as_float((unsigned)(key >> 32)) * 1.0001f + g.slack);
        if (__uint_as_float((unsigned)(key >> 32)) < hit2) bound = -1.f;
      }
      if (xr2 <= bound) {
        key = scan_range<kU>(g, h.z, h.w, qx, qy, qz, key);
        bound = fminf(bound, __uint_as_float((unsigned)(key >> 32)) * 1.0001f + g.slack);
        if (__uint_as_float((unsigned)(key >> 32)) < hit2) bound = -1.f;
      }
    }
    sh.q[0][tid] = qx;
    sh.q[1][tid] = qy;
    sh.q[2][tid] = qz;
    // the eight neighbour rows: wave-uniform steps away from the home row, every one of them inside the array -- all eight 16-byte loads are issued
    // up front behind one wait (a row that does not survive the test below is loaded anyway: no address select)
    asm volatile("" : "+v"(home));                            // (opaque, or the compiler folds the steps back into y and z)
    const float yl2 = ylo * ylo, yh2 = yhi * yhi, zl2 = zlo * zlo, zh2 = zhi * zhi;
    int r_s0[8], r_n[8];
#pragma unroll
    for (int j0 = 0; j0 < 8; j0 += ER_NN_PAD_BATCH) {           // (ER_NN_PAD_BATCH rows' bounds in flight at a time: 4 registers per row)
      i4v rb[ER_NN_PAD_BATCH];
#pragma unroll
      for (int jj = 0; jj < ER_NN_PAD_BATCH; jj++) {
        const int j = j0 + jj, pass = j < 4 ? j : j + 1;
        const int dy = pass % 3 - 1, dz = pass / 3 - 1;
        rb[jj] = *(const ER_GLOBAL i4v*)(csb + (unsigned)(home + (dz * pny + dy) * pnx) * 4u);
      }
#pragma unroll
      for (int jj = 0; jj < ER_NN_PAD_BATCH; jj++) {
        const int j = j0 + jj, pass = j < 4 ? j : j + 1;
        const int dy = pass % 3 - 1, dz = pass / 3 - 1;
        const float e2 = (dy < 0 ? yl2 : (dy > 0 ? yh2 : 0.f)) + (dz < 0 ? zl2 : (dz > 0 ? zh2 : 0.f));
        const bool wl = xl2 + e2 <= bound, wr = xr2 + e2 <= bound;
        const int s0 = wl ? rb[jj].x : rb[jj].y, s1 = wr ? rb[jj].w : rb[jj].z;
        r_s0[j] = s0;
        r_n[j] = e2 <= bound ? s1 - s0 : 0;
        asm volatile("" : "+v"(r_s0[j]), "+v"(r_n[j]));         // materialised HERE: the four bounds die now (left alone, the compiler sinks the selects
      }                                                        // to each row's push and keeps all 32 bound registers alive across the fallback scans)
      __builtin_amdgcn_sched_barrier(0);                       // (... and the scheduler hoists all eight loads above the first batch's selects)
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int n = r_n[j];
      if (n > 0) {
        const int t = n < (1 << 23) ? atomicAdd(&sh.ntask, 1) : kTaskCap;
        if (t < kTaskCap) {
          sh.task_s0[t] = r_s0[j];
          sh.task_nq[t] = (n << 8) | tid;
        } else {                                              // the task list is full (or the range does not fit the packing): scan it here
          key = scan_range<kU>(g, r_s0[j], r_s0[j] + n, qx, qy, qz, key);
        }
      }
    }
  }
  sh.best[tid] = key;
  __syncthreads();
  const int nt = min(sh.ntask, kTaskCap);
  for (int t = tid; t < nt; t += kBlock) {
    const int s0 = sh.task_s0[t], nq = sh.task_nq[t], q = nq & 255;
    const unsigned long long k = scan_range<kU>(g, s0, s0 + (nq >> 8), sh.q[0][q], sh.q[1][q], sh.q[2][q], kNoHit);
    atomicMin(&sh.best[q], k);
  }
  __syncthreads();
  key = sh.best[tid];
  best_d = __uint_as_float((unsigned)(key >> 32));
  return (int)(unsigned)(key & 0xffffffffull);              // 0xffffffff -> -1
}

using NnSh = NnShared;
constexpr int kPrecheckUnroll = 3;   // candidates per trip of the any-hit pre-check (see nn_block)

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

// RansacCurvature::getFitness (GlobalRegistration/RansacCurvature.h:661-704) for MANY pose hypotheses of one
// (source, target) pair: blockIdx.y = hypothesis, blockIdx.x strides over the source points.  Float32 transform
// ([PCL] transformPointCloud with a Matrix4f), exact NN, inlier iff d < threshold^2 (float compare, :670,:687);
// per hypothesis the inlier count (exact) and the float64 sum of the inlier distances.
__global__ __launch_bounds__(kBlock) void k_ransac_fitness(const float4* __restrict__ src_sorted, int n, const float* __restrict__ hyp,
                                                           int hyp0, Grid g, float radius, float max_range,
                                                           int* __restrict__ count, double* __restrict__ sum) {
  __shared__ NnSh sh;
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

// ---- pair groups -------------------------------------------------------------------------------------------------------
// The reference's two loops run over a LIST of fragment pairs (CorresApp.cpp:121,220: #pragma omp parallel for).  Here a whole
// group of pairs goes through every stage in ONE launch: blockIdx.y = pair, blockIdx.x = a slice of that pair's source points.
// A 250 k-point pair alone is ~1000 workgroups of latency-bound exact-NN work -- half of what the 256 CUs hold -- and every
// launch, memset and small copy costs 5-20 us of an otherwise idle stream; 50 pairs per launch fill the chip and turn the
// host side into a handful of calls per stage (round 2: ~30 launches and ~7 copies PER PAIR over four streams).
// PairDev = what the kernels need to know about one pair of the group; read with scalar loads (blockIdx.y is wave-uniform).
struct PairDev {
  const float4* src_sorted;      // source, cell-sorted {x, y, z, original index}
  const float* src_xyz;          // source, file order (information matrix, CorresApp.cpp:192-196)
  const float* src_nrm;
  const float* tgt_xyz;
  const float* tgt_nrm;
  const float4* tgt_xn;          // target, file order, {xyz | normal} records of 32 bytes
  Grid g;                        // the target's uniform grid
  Mat12d T;                      // transform of the pre-check / FindCorrespondence (Matrix4d, rows 0..2)
  float* X;                      // [3 n]   ICP: the source as the loop transforms it (cell-sorted order)
  int* match;                    // [n]     FindCorrespondence: NN index or -1, file order
  int* block_count;              // [nb]
  int* block_offset;             // [nb]
  int* pairs;                    // [2 n]   compacted (target index, source index) list
  double* partial;               // [nbi][32] per-workgroup sums of one ICP iteration
  int n, nb, nbi, pts;           // source points, ceil(n / 256); nbi, pts: unused since round 4 (the points per thread of k_icp_iter are a launch argument)
  double fx_scale[32];           // round 6: power-of-two scale of each of the 29 ICP sums (see k_icp_iter: fixed-point partial sums) ...
  double fx_inv[32];             // ... and its reciprocal
};

// ---- the ICP loop's state lives on the device ------------------------------------------------------------------------
// pcl::IterativeClosestPoint::align's loop variables (final_transformation_, the last increment, the previous MSE, the
// iteration counter and the convergence flags), one per pair of the group.  k_icp_final updates them from the iteration's sums
// -- 6x6 solve, increment, PCL's stop rule -- and k_icp_iter reads the increment (and the `done` flag) from here, so a whole
// chunk of iterations is enqueued without a host round trip; workgroups that find `done` set return at once.
struct IcpDev {
  float fin[16];           // final_transformation_ (= the guess before the first iteration)
  float delta[16];         // last increment (identity before the first solve)
  float prev_delta[16];
  double prev_mse;
  int iter, done, conv, apply;   // apply: the next k_icp_iter multiplies X by delta first
  int init_apply, pad[3];        // iteration 0 reads the source itself and applies the guess unless it is the identity
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


constexpr int kIcpPtsMax = 2;              // slices of 256 source points per workgroup of k_icp_iter.  Round 6: two, not eight -- a workgroup of eight slices lasts
                                           // ~150 us of a ~700 us launch, and the launch ends with its last workgroup: ICP phase of the kinfu-like list 3.97 -> 3.48 ms,
                                           // hard list 5.17 -> 4.93, uniform 2.30 -> 2.21 (1: worse again; profiles/r06n_ab_icp_points_per_workgroup.txt).  Free to
                                           // choose since the sums are order-independent: every digest identical.

// Registration pre-check, CorresApp.cpp:257-264, for a group of pairs: counts[pair] = #{k : d2 < reg_dist^2}.  blockIdx.x
// strides over the pair's points; ONE atomic per workgroup.
__global__ __launch_bounds__(kBlock) void k_count_inliers(const PairDev* __restrict__ P, float radius, double maxd2, int* __restrict__ counts) {
  __shared__ NnSh sh;
  const PairDev& p = P[blockIdx.y];
  const int n = p.n;
  if ((int)blockIdx.x * kBlock >= n) return;                 // (wave-uniform: whole workgroups beyond a short pair's points)
  int local = 0;
  // any-hit threshold: a float strictly below both limits of the count test (d <= radius^2 as doubles, d < maxd2), so "d < hit2" implies the test
  const double lim = fmin((double)radius * (double)radius, maxd2);
  float hit2 = (float)lim;
  if ((double)hit2 >= lim) hit2 = __uint_as_float(__float_as_uint(hit2) - 1u);      // (lim > 0: the next float below)
  for (int base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {
    const int k = base + (int)threadIdx.x;
    float qx = 0.f, qy = 0.f, qz = 0.f, d;
    if (k < n) {
      const float4 s = p.src_sorted[k];
      xform_d(p.T, s.x, s.y, s.z, qx, qy, qz);
    }
    const int i = nn_block<kPrecheckUnroll>(sh, p.g, k < n, qx, qy, qz, radius * radius, d, hit2);
    if (k < n && i >= 0 && (double)d <= (double)radius * (double)radius && (double)d < maxd2) local++;
  }
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off);
  __shared__ int part[kBlock / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int w = 0; w < kBlock / 64; w++) s += part[w];
    if (s) atomicAdd(&counts[blockIdx.y], s);
  }
}

// v (finite, |v| < 2^126) as a signed 128-bit integer, truncated towards zero below the units (hi: signed high word, lo: low word, two's complement).
__device__ __forceinline__ void to_fixed128(double v, long long& hi, unsigned long long& lo) {
  hi = 0;
  lo = 0;
  const double a = fabs(v);
  if (!(a > 0.0) || !(a < 8.0e37)) return;                       // 0, NaN, inf (cannot happen under the host's bound)
  int e;
  const double f = frexp(a, &e);                                 // a = f 2^e, 0.5 <= f < 1
  const unsigned long long m = (unsigned long long)ldexp(f, 53); // 2^52 <= m < 2^53, exact
  const int sh = e - 53;                                         // a = m 2^sh
  unsigned long long mh = 0, ml = 0;
  if (sh >= 64) mh = m << (sh - 64);
  else if (sh > 0) { mh = m >> (64 - sh); ml = m << sh; }
  else if (sh > -64) ml = m >> (-sh);
  if (v < 0.0) {                                                 // negate
    ml = ~ml + 1ull;
    mh = ~mh + (ml == 0ull ? 1ull : 0ull);
  }
  hi = (long long)mh;
  lo = ml;
}

// One ICP iteration of every ACTIVE pair of the group in one launch (blockIdx.y -> active[y] = the pair's slot):
//   iteration 0: X <- guess * source (IterativeClosestPoint::transformCloud, float32; a plain copy for an identity guess);
//   later:       X <- delta * X (the previous iteration's increment, float32);
//   correspondence estimation (exact NN, kept if d^2 <= max_dist^2); the sums of TransformationEstimationPointToPlaneLLS over
//   the kept correspondences: acc[0..20] upper triangle of AtA row by row, [21..26] Atb, [27] sum of d^2, [28] count.
// A thread takes `pts` points (slices of kBlock consecutive points: the NN search is a workgroup-cooperative step per slice)
// and adds their rows up privately; then thread rows -> wave shuffle tree -> LDS -> ONE partial vector per workgroup in
// `partial`; k_icp_final adds the partial vectors in a fixed order.  No float64 atomics: the sums are bit-reproducible from
// run to run (they still differ from a sequential CPU sum in the last bits).
__global__ __launch_bounds__(kBlock, 6) void k_icp_iter(const PairDev* __restrict__ P, const int* __restrict__ active, const IcpDev* __restrict__ S,
                                                     float radius, double maxd2, int pts) {
  __shared__ NnSh sh;
  const int slot = active[blockIdx.y];
  const PairDev& p = P[slot];
  const IcpDev& st = S[slot];
  if (st.done || (int)blockIdx.x >= (p.nb + pts - 1) / pts) return;   // the loop has ended / a shorter pair: wave-uniform exits
  const int n = p.n;
  const bool first = st.iter == 0;
  const int apply = first ? st.init_apply : st.apply;        // wave-uniform (scalar loads)
  const float* __restrict__ dm = first ? st.fin : st.delta;
  float* __restrict__ X = p.X;
  // Per slice of 256 points: NN, the point's 29 row values, a wave-level halving butterfly over the 32 value slots (29 used) -- at each
  // step the upper half of the lanes keeps the upper half of the remaining slots and hands the lower half over (and vice versa):
  // 16 + 8 + 4 + 2 + 1 exchanges and a final pair add instead of 29 x 6 shuffle-adds; lane l ends up with the wave total of slot
  // l >> 1 -- and the wave adds it to ITS row of the LDS accumulator.  The row values are live only between the NN search and the
  // butterfly, so the kernel keeps the register footprint of the plain NN kernels (occupancy is what the latency-bound search
  // needs: with thread-private float64 sums carried across the slices the kernel held 126 VGPRs = 4 waves per SIMD).
  // The sums ACROSS waves are exact.  A wave's 29 totals of one slice (float64, butterfly order: a function of the slice's 64 points alone) are converted
  // to 128-bit fixed point -- a power-of-two scale per sum, sized on the host from certain bounds (icp_fixed_point_scales: the total stays below 2^121);
  // with 120 bits the conversion is exact for anything but denormal dust -- and from there on everything is integer addition: associative, so a
  // pair's sums, its transform and its iteration count do not depend on `pts`, the chunking of the loop, the list the pair is part of or ER_ICP_SHARES.
  // (64 bits were not enough: a scale sized for the worst case leaves the point-to-plane residual sums A^T b -- terms of 1e-3 that cancel -- a
  // quantisation noise 300 x float64's, and on ill-posed pairs, where the solve divides noise by noise, 385 of configs[4]'s 2367 accepted pairs then
  // wandered for all 20 iterations instead of 12: profiles/r06g_allpairs_fixed_point_width.txt.)
  // The conversion waits until the slices are done: the wave totals of every slice are parked in LDS (8 KB) and converted after the loop, where the
  // registers of the search are free (inside the loop the 128-bit shifts pushed the kernel 12 bytes over its 80-VGPR budget).
  __shared__ double dslice[kIcpPtsMax][kBlock / 64][32];
  __shared__ unsigned long long part_lo[kBlock / 32][32];
  __shared__ long long part_hi[kBlock / 32][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = threadIdx.x; t < kIcpPtsMax * (kBlock / 64) * 32; t += kBlock) (&dslice[0][0][0])[t] = 0.0;   // (the first barrier of nn_block orders this)
  __shared__ double sfx[32];                                 // the scales wait in LDS: two registers less across the search (the kernel sits on its 80)
  if (threadIdx.x < 32) sfx[threadIdx.x] = p.fx_scale[threadIdx.x];
  for (int c = 0; c < pts; c++) {
    const int k = (blockIdx.x * pts + c) * kBlock + threadIdx.x;
    if ((blockIdx.x * pts + c) * kBlock >= n) break;         // wave-uniform
    float sx = 0.f, sy = 0.f, sz = 0.f;
    if (k < n) {
      if (first) {
        const float4 s = p.src_sorted[k];
        sx = s.x, sy = s.y, sz = s.z;
      } else {
        const ER_GLOBAL float* Xg = (const ER_GLOBAL float*)((const ER_GLOBAL char*)X + (unsigned)k * 12u);
        sx = Xg[0], sy = Xg[1], sz = Xg[2];
      }
      if (apply) {
        const float x = sx, y = sy, z = sz;
        sx = ((dm[0] * x + dm[1] * y) + dm[2] * z) + dm[3];
        sy = ((dm[4] * x + dm[5] * y) + dm[6] * z) + dm[7];
        sz = ((dm[8] * x + dm[9] * y) + dm[10] * z) + dm[11];
      }
      if (apply || first) {
        ER_GLOBAL float* Xw = (ER_GLOBAL float*)((ER_GLOBAL char*)X + (unsigned)k * 12u);
        Xw[0] = sx;
        Xw[1] = sy;
        Xw[2] = sz;
      }
    }
    float d;
    const int i = nn_block(sh, p.g, k < n, sx, sy, sz, radius * radius, d);
    double w[32];
#pragma unroll
    for (int t = 0; t < 32; t++) w[t] = 0.0;
    if (k < n && i >= 0 && (double)d <= (double)radius * (double)radius && !((double)d > maxd2)) {
      const ER_GLOBAL f4v* rec = (const ER_GLOBAL f4v*)((const ER_GLOBAL char*)p.tgt_xn + (unsigned)i * 32u);
      const f4v tp = rec[0], tn = rec[1];
      const float dx = tp.x, dy = tp.y, dz = tp.z, nx = tn.x, ny = tn.y, nz = tn.z;
      w[27] = (double)d;
      w[28] = 1.0;
      if (isfinite(sx) && isfinite(sy) && isfinite(sz) && isfinite(nx) && isfinite(ny) && isfinite(nz)) {
        const double a = (double)(nz * sy - ny * sz);      // float32 expressions widened to double (PCL)
        const double b = (double)(nx * sz - nz * sx);
        const double cc = (double)(ny * sx - nx * sy);
        const double dnx = nx, dny = ny, dnz = nz;
        w[0] = a * a;  w[1] = a * b;  w[2] = a * cc;  w[3] = a * dnx;  w[4] = a * dny;  w[5] = a * dnz;
        w[6] = b * b;  w[7] = b * cc;  w[8] = b * dnx; w[9] = b * dny; w[10] = b * dnz;
        w[11] = cc * cc; w[12] = cc * dnx; w[13] = cc * dny; w[14] = cc * dnz;
        w[15] = dnx * dnx; w[16] = dnx * dny; w[17] = dnx * dnz;
        w[18] = dny * dny; w[19] = dny * dnz;
        w[20] = dnz * dnz;
        const double e = (double)(nx * dx + ny * dy + nz * dz - nx * sx - ny * sy - nz * sz);
        w[21] = a * e; w[22] = b * e; w[23] = cc * e; w[24] = dnx * e; w[25] = dny * e; w[26] = dnz * e;
      }
    }
#pragma unroll
    for (int half = 16, mask = 32; half >= 1; half >>= 1, mask >>= 1) {
      const bool upper = (lane & mask) != 0;
#pragma unroll
      for (int t = 0; t < half; t++) {
        const double send = upper ? w[t] : w[t + half];
        const double keep = upper ? w[t + half] : w[t];
        w[t] = keep + __shfl_xor(send, mask);
      }
    }
    w[0] += __shfl_xor(w[0], 1);
    if ((lane & 1) == 0) dslice[c][wave][lane >> 1] = w[0];
  }
  __syncthreads();
  {
    const int slot = threadIdx.x & 31, g = threadIdx.x >> 5;     // 8 groups of 32 threads: group g converts slice g of this workgroup
    unsigned long long lo = 0;
    long long hi = 0;
    if (slot < 29 && g < pts) {
      const double fx = sfx[slot];
#pragma unroll
      for (int w2 = 0; w2 < kBlock / 64; w2++) {
        long long qh;
        unsigned long long ql;
        to_fixed128(dslice[g][w2][slot] * fx, qh, ql);           // (a power-of-two scale: the product is exact)
        const unsigned long long nlo = lo + ql;
        hi += qh + (long long)(nlo < lo);
        lo = nlo;
      }
    }
    part_lo[g][slot] = lo;
    part_hi[g][slot] = hi;
  }
  __syncthreads();
  if (threadIdx.x < 29) {
    unsigned long long lo = 0;
    long long hi = 0;
#pragma unroll
    for (int g = 0; g < kBlock / 32; g++) {
      const unsigned long long nlo = lo + part_lo[g][threadIdx.x];
      hi += part_hi[g][threadIdx.x] + (long long)(nlo < lo);
      lo = nlo;
    }
    unsigned long long* out = reinterpret_cast<unsigned long long*>(p.partial) + ((size_t)blockIdx.x * 32 + threadIdx.x) * 2;
    out[0] = lo;
    out[1] = (unsigned long long)hi;
  }
}

// Second half of the reduction, one workgroup PER ACTIVE PAIR: adds the pair's per-workgroup partial vectors in a fixed order
// (8 strided slices per value, then the slices in order) and decides -- 6x6 solve, increment, stop rule: the loop never leaves
// the device.  A separate launch on purpose: finishing inside k_icp_iter ("last workgroup done" ticket + __threadfence) costs
// an L2 write-back per workgroup on this multi-XCD part (measured in round 1: 100-600 us per launch instead of ~25).
__global__ __launch_bounds__(kBlock) void k_icp_final(const PairDev* __restrict__ P, const int* __restrict__ active, IcpDev* __restrict__ S, IcpParams prm,
                                                      int pts) {
  const int slot = active[blockIdx.x];
  IcpDev* st = S + slot;
  if (st->done) return;
  const unsigned long long* __restrict__ partial = reinterpret_cast<const unsigned long long*>(P[slot].partial);
  const int nparts = (P[slot].nb + pts - 1) / pts;
  const int val = threadIdx.x & 31, slice = threadIdx.x >> 5;   // 32 x 8
  unsigned long long lo = 0;                                    // (128-bit integer sums: any order gives the same total)
  long long hi = 0;
  if (val < 29) {
    for (int b = slice; b < nparts; b += 8) {
      const unsigned long long nlo = lo + partial[((size_t)b * 32 + val) * 2];
      hi += (long long)partial[((size_t)b * 32 + val) * 2 + 1] + (long long)(nlo < lo);
      lo = nlo;
    }
  }
  __shared__ unsigned long long fin_lo[8][32];
  __shared__ long long fin_hi[8][32];
  fin_lo[slice][val] = lo;
  fin_hi[slice][val] = hi;
  __syncthreads();
  __shared__ double tot[32];
  if (threadIdx.x < 29) {
    unsigned long long l = 0;
    long long h = 0;
#pragma unroll
    for (int sl = 0; sl < 8; sl++) {
      const unsigned long long nl = l + fin_lo[sl][threadIdx.x];
      h += fin_hi[sl][threadIdx.x] + (long long)(nl < l);
      l = nl;
    }
    // value = h 2^64 + l (two's complement): two roundings to float64, far below the wave sums' own
    tot[threadIdx.x] = (ldexp((double)h, 64) + (double)l) * P[slot].fx_inv[threadIdx.x];
  }
  __syncthreads();
  if (threadIdx.x == 0) dev_icp_decide(tot, st, prm);
}

// getFitnessScore-style diagnostic (logging only, CorresApp.cpp:307): per pair the sum and the number of the squared NN
// distances of final * source inside the search radius -> fit[2 pair], fit[2 pair + 1].
__global__ __launch_bounds__(kBlock) void k_fitness(const PairDev* __restrict__ P, const IcpDev* __restrict__ S, float radius, double* __restrict__ fit) {
  __shared__ NnSh sh;
  const PairDev& p = P[blockIdx.y];
  const int n = p.n;
  if ((int)blockIdx.x * kBlock >= n) return;
  const float* __restrict__ M = S[blockIdx.y].fin;
  double v[2] = {0.0, 0.0};
  for (int base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {
    const int k = base + (int)threadIdx.x;
    float qx = 0.f, qy = 0.f, qz = 0.f, d;
    if (k < n) {
      const float4 s = p.src_sorted[k];
      qx = ((M[0] * s.x + M[1] * s.y) + M[2] * s.z) + M[3];
      qy = ((M[4] * s.x + M[5] * s.y) + M[6] * s.z) + M[7];
      qz = ((M[8] * s.x + M[9] * s.y) + M[10] * s.z) + M[11];
    }
    const int i = nn_block(sh, p.g, k < n, qx, qy, qz, radius * radius, d);
    if (k < n && i >= 0 && (double)d <= (double)radius * (double)radius) {
      v[0] += (double)d;
      v[1] += 1.0;
    }
  }
  __syncthreads();
  block_reduce_atomic<2>(v, fit + 2 * (size_t)blockIdx.y);
}

// FindCorrespondence, CorresApp.cpp:144-161: match[original index] = NN index passing the distance and normal tests, else -1.
__global__ __launch_bounds__(kBlock) void k_find_corr(const PairDev* __restrict__ P, float radius, double dist2, double normal_cos) {
  __shared__ NnSh sh;
  const PairDev& p = P[blockIdx.y];
  const int n = p.n;
  if ((int)blockIdx.x >= p.nb) return;
  const int q = blockIdx.x * kBlock + threadIdx.x;             // q = position in the source's cell-sorted order
  float qx = 0.f, qy = 0.f, qz = 0.f, d;
  int k = 0;
  if (q < n) {
    const float4 s = p.src_sorted[q];
    k = __float_as_int(s.w);                                   // original (file-order) index of this source point
    xform_d(p.T, s.x, s.y, s.z, qx, qy, qz);
  }
  const int i = nn_block(sh, p.g, q < n, qx, qy, qz, radius * radius, d);
  if (q >= n) return;
  int m = -1;
  if (i >= 0 && (double)d <= (double)radius * (double)radius && (double)d < dist2) {         // :154
    const double nx = p.src_nrm[3 * k], ny = p.src_nrm[3 * k + 1], nz = p.src_nrm[3 * k + 2];
    const float tnx = (float)((p.T.m[0] * nx + p.T.m[1] * ny) + p.T.m[2] * nz);              // n' = R n (double -> float)
    const float tny = (float)((p.T.m[4] * nx + p.T.m[5] * ny) + p.T.m[6] * nz);
    const float tnz = (float)((p.T.m[8] * nx + p.T.m[9] * ny) + p.T.m[10] * nz);
    // NormalDot, CorresApp.h:58-60: float32 products/sums, compared as double
    const float4 tn = p.tgt_xn[2 * (size_t)i + 1];                                              // the matched target normal: one aligned 16-byte load
    const float dot = (tn.x * tnx + tn.y * tny) + tn.z * tnz;
    if ((double)dot > normal_cos) m = i;                                                       // :155
  }
  p.match[k] = m;
}

// The accepted RANSAC hypothesis once more, this time keeping the lists (RansacCurvature.h:661-704: `inliers`, `inliers_target`):
// match[original source index] = NN index if d < threshold^2, else -1; acc[20] += d over the inliers.  (One pair: P[0].)
__global__ __launch_bounds__(kBlock) void k_ransac_match(const PairDev* __restrict__ P, Mat12f M, float radius, float max_range, double* __restrict__ acc) {
  __shared__ NnSh sh;
  const PairDev& p = P[0];
  const int n = p.n;
  const int q = blockIdx.x * kBlock + threadIdx.x;
  float qx = 0.f, qy = 0.f, qz = 0.f, d;
  int k = 0;
  if (q < n) {
    const float4 s = p.src_sorted[q];
    k = __float_as_int(s.w);
    qx = ((M.m[0] * s.x + M.m[1] * s.y) + M.m[2] * s.z) + M.m[3];
    qy = ((M.m[4] * s.x + M.m[5] * s.y) + M.m[6] * s.z) + M.m[7];
    qz = ((M.m[8] * s.x + M.m[9] * s.y) + M.m[10] * s.z) + M.m[11];
  }
  const int i = nn_block(sh, p.g, q < n, qx, qy, qz, radius * radius, d);
  const bool hit = q < n && i >= 0 && d < max_range;
  if (q < n) p.match[k] = hit ? i : -1;
  double v[1] = {hit ? (double)d : 0.0};
  __syncthreads();
  block_reduce_atomic<1>(v, acc + 20);
}

// Matches per block of 256 consecutive ORIGINAL indices (the order of the output list) and, optionally, the
// information matrix of CorresApp.cpp:186-208 over the matched, UNtransformed source points (target points: RansacCurvature.h
// :723-731 with which = 1):  info[0..2] = sum 2sx,2sy,2sz; [3..5] = sum (4sz^2+4sy^2),(4sz^2+4sx^2),(4sy^2+4sx^2);
// [6..8] = sum -4sysx, -4szsx, -4szsy; [9] = count   (the distinct terms of sum A^T A, A = [I | 2*skew-like(s)]); info is
// kAcc doubles per pair (source terms at 0, target terms at 10).
// A workgroup takes kCountSlices consecutive blocks, keeps the twenty sums thread-private across them and leaves ONE partial vector per workgroup in the
// pair's `partial` scratch (free here: the ICP loop that owns it has ended); k_scan_blocks adds the partial vectors in a fixed order: no float64
// atomics, a result that does not depend on arrival order.
constexpr int kCountSlices = 8;
__global__ __launch_bounds__(kBlock) void k_count_blocks(const PairDev* __restrict__ P, int want_source, int want_target) {
  const PairDev& p = P[blockIdx.y];
  const int n = p.n, nb = p.nb;
  const int b0 = blockIdx.x * kCountSlices;
  if (b0 >= nb) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __shared__ int wcnt[kCountSlices][kBlock / 64];
  double v[20];
#pragma unroll
  for (int i = 0; i < 20; i++) v[i] = 0.0;
#pragma unroll
  for (int c = 0; c < kCountSlices; c++) {
    const int k = (b0 + c) * kBlock + (int)threadIdx.x;
    const int m = (b0 + c < nb && k < n) ? p.match[k] : -1;
    const bool hit = m >= 0;
    const unsigned long long b = __ballot(hit);
    if (lane == 0) wcnt[c][wave] = __popcll(b);
    if (hit) {                                                                                   // :192-204
#pragma unroll
      for (int which = 0; which < 2; which++) {
        if (!(which ? want_target : want_source)) continue;      // uniform
        const float* __restrict__ s = which ? p.tgt_xyz + 3 * (size_t)m : p.src_xyz + 3 * (size_t)k;
        const float sx = s[0], sy = s[1], sz = s[2];
        const double ax = (double)(2 * sx), ay = (double)(2 * sy), az = (double)(2 * sz);
        double* u = v + 10 * which;
        u[0] += ax; u[1] += ay; u[2] += az;
        u[3] += az * az + ay * ay;     // (0*0 + (-2sz)(-2sz)) + (2sy)(2sy)
        u[4] += az * az + ax * ax;     // ((2sz)(2sz) + 0*0) + (-2sx)(-2sx)
        u[5] += ay * ay + ax * ax;     // ((-2sy)(-2sy) + (2sx)(2sx)) + 0*0
        u[6] += ay * (-ax);            // (3,4): (2sy)(-2sx)
        u[7] += (-az) * ax;            // (3,5): (-2sz)(2sx)
        u[8] += az * (-ay);            // (4,5): (2sz)(-2sy)
        u[9] += 1.0;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < kCountSlices && b0 + (int)threadIdx.x < nb)
    p.block_count[b0 + threadIdx.x] = ((wcnt[threadIdx.x][0] + wcnt[threadIdx.x][1]) + wcnt[threadIdx.x][2]) + wcnt[threadIdx.x][3];
  if (!(want_source | want_target)) return;                    // uniform
  __shared__ double red[kBlock / 64][20];
#pragma unroll
  for (int i = 0; i < 20; i++) {
    double t = v[i];
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
    if (lane == 0) red[wave][i] = t;
  }
  __syncthreads();
  if (threadIdx.x < 20) p.partial[(size_t)blockIdx.x * 32 + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// Exclusive scan of a pair's per-block match counts (a few thousand blocks at most) and, with want_info, the fixed-order total of the information
// partial vectors k_count_blocks left (info[0..19] of the pair are OVERWRITTEN; info[20], the fitness sum of k_ransac_match, is left alone): one workgroup per pair.
__global__ __launch_bounds__(1024) void k_scan_blocks(const PairDev* __restrict__ P, int* __restrict__ totals, double* __restrict__ info, int want_info) {
  __shared__ int buf[1024];
  __shared__ int carry;
  const PairDev& p = P[blockIdx.x];
  const int nb = p.nb;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + (int)threadIdx.x;
    const int v = i < nb ? p.block_count[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int t = threadIdx.x >= (unsigned)off ? buf[threadIdx.x - off] : 0;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) p.block_offset[i] = carry + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += buf[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry;
  if (want_info) {                                              // (uniform) 32 values x 32 strided slices of the partial vectors, then the slices in order
    __shared__ double fin[32][33];
    const int val = threadIdx.x & 31, slice = threadIdx.x >> 5;
    const int nparts = (nb + kCountSlices - 1) / kCountSlices;
    double q = 0.0;
    if (val < 20)
      for (int b = slice; b < nparts; b += 32) q += p.partial[(size_t)b * 32 + val];
    fin[slice][val] = q;
    __syncthreads();
    if (threadIdx.x < 20) {
      double r = 0.0;
#pragma unroll
      for (int sl = 0; sl < 32; sl++) r += fin[sl][threadIdx.x];
      info[(size_t)blockIdx.x * kAcc + threadIdx.x] = r;
    }
  }
}

// Stable compaction: pairs (match[k], k) in ascending k (CorresApp.cpp:157, file order of corres_*.txt).  p.pairs may be the caller's own buffer --
// device memory or page-locked host memory (er_find_correspondence_batch) -- in which case the list is complete where it is wanted when the kernel ends.
// (Staging the block's matches in LDS and storing them as aligned 16-byte bursts changed nothing for a host destination: profiles/r06l_fc_modes.txt.)
__global__ __launch_bounds__(kBlock) void k_compact(const PairDev* __restrict__ P) {
  const PairDev& p = P[blockIdx.y];
  const int n = p.n;
  if ((int)blockIdx.x >= p.nb) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = k < n ? p.match[k] : -1;
  const unsigned long long b = __ballot(m >= 0);
  __shared__ int wcnt[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wcnt[wave] = __popcll(b);
  __syncthreads();
  if (m >= 0) {
    int o = p.block_offset[blockIdx.x] + __popcll(b & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; w++) o += wcnt[w];
    reinterpret_cast<int2*>(p.pairs)[o] = make_int2(m, k);       // (o < n: every match has its own k; the buffer is 8-byte aligned)
  }
}

// ---- uniform grids of a CHUNK of clouds, built on the device ---------------------------------------------------------------
// (the reference builds a kd-tree per pair and per function, CorresApp.cpp:129,238; here once per fragment)
// Up to kCloudChunk clouds go through ONE set of launches (round 4; before, every cloud had its own ~23 launches and a list of fragments was bound by
// the host's launch rate, not by PCIe: profiles/r04y_cloud_build_timeline.txt):
//   k_chunk_bounds  per cloud (blockIdx.y): min / max of the coordinates (float bits mapped to ordered ints, block reduce, 7 atomics per
//                   workgroup) + a non-finite flag;
//   k_chunk_cells   cell id of every point (the same float32 expression nn_block evaluates for a query) -> key = cloud << shift | cell, the
//                   histogram of every cloud's cells in one concatenated array [ncell_0 + 1 | ncell_1 + 1 | ...], and the interleaved
//                   {point, normal} records;
//   hipcub          ONE stable radix sort of (key, position in the chunk) and ONE prefix sum over the concatenated histograms: element 0 of
//                   cloud k's segment holds -n_(k-1), which cancels the running total at the segment's start, so every segment comes out as that
//                   cloud's own cell_start (0 ... n_k);
//   k_chunk_gather  sorted[s] = {x, y, z, original index within its cloud}; the clouds' sorted arrays are consecutive pieces of one array.
// A stable sort keeps the points of a cell in file order, like a counting sort on the host would: the layout -- and with it the
// order of every float64 reduction that walks the cloud -- is reproducible from run to run, and the same for a cloud built alone or in a list.
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

constexpr int kCloudChunk = 8;          // clouds per chunk (3 key bits above the cell id)
struct GridDims {
  float org[3];
  float cell;
  int dim[3];
};
struct ChunkDesc {                      // passed by value (kernel argument)
  int m;                                // clouds in the chunk
  int shift;                            // key = cloud << shift | cell
  int n[kCloudChunk];
  int pt_off[kCloudChunk + 1];          // prefix sum of n: a point's position in the chunk
  long cs_off[kCloudChunk + 1];         // prefix sum of (cells + 1): where a cloud's cell_start begins in the chunk's array
  const float* xyz[kCloudChunk];
  const float* nrm[kCloudChunk];
  float4* xn[kCloudChunk];
  GridDims G[kCloudChunk];
};

__global__ __launch_bounds__(kBlock) void k_chunk_bounds(ChunkDesc D, int* __restrict__ out8) {
  const int y = blockIdx.y, n = D.n[y];
  const float* __restrict__ xyz = D.xyz[y];
  int* out7 = out8 + 8 * y;
  int lo[3] = {INT_MAX, INT_MAX, INT_MAX}, hi[3] = {INT_MIN, INT_MIN, INT_MIN};
  int bad = 0, nmax = 0;                                        // nmax: bits of the largest finite |normal component| (both uploads are in: stage A)
  const float* __restrict__ nrm = D.nrm[y];
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const float v = xyz[3 * (size_t)i + a];
      bad |= !isfinite(v);
      const int o = ordered_int(v);
      lo[a] = min(lo[a], o);
      hi[a] = max(hi[a], o);
      const float nv = fabsf(nrm[3 * (size_t)i + a]);
      if (isfinite(nv)) nmax = max(nmax, __float_as_int(nv));
    }
  }
  for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, __shfl_down(nmax, off));
  if ((threadIdx.x & 63) == 0 && nmax > 0) atomicMax(&out8[8 * y + 7], nmax);
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
  if (threadIdx.x < 7 && (int)blockIdx.x * kBlock < n) {       // 7 atomics per workgroup that saw points
    int v = part[0][threadIdx.x];
    for (int w = 1; w < kBlock / 64; w++)
      v = threadIdx.x < 3 ? min(v, part[w][threadIdx.x]) : (threadIdx.x < 6 ? max(v, part[w][threadIdx.x]) : (v | part[w][threadIdx.x]));
    if (threadIdx.x < 3) atomicMin(&out7[threadIdx.x], v);
    else if (threadIdx.x < 6) atomicMax(&out7[threadIdx.x], v);
    else if (v) atomicOr(&out7[6], 1);
  }
}

// xn[2 i] = {x, y, z, 0}, xn[2 i + 1] = {nx, ny, nz, 0} in file order: the matched target point of k_icp_iter is one aligned
// 32-byte record = one cache line instead of two 12-byte gathers from two arrays (those were ~30 % of an ICP iteration).
__global__ __launch_bounds__(kBlock) void k_chunk_cells(ChunkDesc D, unsigned* __restrict__ key, unsigned* __restrict__ idx, int* __restrict__ count) {
  const int y = blockIdx.y, n = D.n[y];
  int* __restrict__ cnt = count + D.cs_off[y];
  if (blockIdx.x == 0 && threadIdx.x == 0) cnt[0] = y > 0 ? -D.n[y - 1] : 0;   // (see the prefix sum above; nothing else touches element 0)
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const float* __restrict__ xyz = D.xyz[y];
  const float* __restrict__ nrm = D.nrm[y];
  const GridDims G = D.G[y];
  const float p[3] = {xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]};
  int q[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    q[a] = (int)floorf((p[a] - G.org[a]) / G.cell);
    q[a] = min(max(q[a], 0), G.dim[a] - 1);
  }
  const int c = ((q[2] + 2) * (G.dim[1] + 4) + (q[1] + 2)) * (G.dim[0] + 4) + (q[0] + 2);   // two rings of empty cells around the box (struct Grid)
  const unsigned g = (unsigned)(D.pt_off[y] + i);
  key[g] = ((unsigned)y << D.shift) | (unsigned)c;
  idx[g] = g;
  atomicAdd(&cnt[c + 1], 1);                                 // integer histogram: the result does not depend on the order
  float4* __restrict__ xn = D.xn[y];
  xn[2 * (size_t)i] = make_float4(p[0], p[1], p[2], 0.f);
  xn[2 * (size_t)i + 1] = make_float4(nrm[3 * (size_t)i], nrm[3 * (size_t)i + 1], nrm[3 * (size_t)i + 2], 0.f);
}

__global__ __launch_bounds__(kBlock) void k_chunk_gather(ChunkDesc D, const unsigned* __restrict__ key, const unsigned* __restrict__ idx, int total,
                                                         float4* __restrict__ sorted) {
  const int s = blockIdx.x * kBlock + threadIdx.x;
  if (s >= total) return;
  const int y = (int)(key[s] >> D.shift);
  const int i = (int)idx[s] - D.pt_off[y];
  const float* __restrict__ xyz = D.xyz[y];
  float4* __restrict__ out = sorted + (size_t)s + (size_t)kSentinel * (size_t)y;       // kSentinel entries of +inf behind every cloud (scan_range)
  *out = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __int_as_float(i));
  if (s + 1 == D.pt_off[y + 1]) {                                // the cloud's last point in sorted order also writes the sentinels
    const float inf = __int_as_float(0x7f800000);
    for (int k = 1; k <= kSentinel; k++) out[k] = make_float4(inf, inf, inf, __int_as_float(-1));
  }
}

// occ[c] = 1 iff some cell of c's 3 x 3 x 3 neighbourhood holds a point (c over the padded array; cells of the outermost ring get 0: no query is searched
// from there).  One thread per cell, nine row sums of three cells each from the finished cell_start.
__global__ __launch_bounds__(kBlock) void k_chunk_occ(ChunkDesc D, const int* __restrict__ cell_start, unsigned char* __restrict__ occ) {
  const int y = blockIdx.y;
  const GridDims G = D.G[y];
  const int pnx = G.dim[0] + 4, pny = G.dim[1] + 4, pnz = G.dim[2] + 4;
  const long ncell = (long)pnx * pny * pnz;
  const long c = (long)blockIdx.x * kBlock + threadIdx.x;
  if (c >= ncell) return;
  const int* __restrict__ cs = cell_start + D.cs_off[y];
  const int x = (int)(c % pnx), yy = (int)((c / pnx) % pny), z = (int)(c / ((long)pnx * pny));
  int any = 0;
  if (x >= 1 && x <= pnx - 2 && yy >= 1 && yy <= pny - 2 && z >= 1 && z <= pnz - 2) {
#pragma unroll
    for (int dz = -1; dz <= 1; dz++)
#pragma unroll
      for (int dy = -1; dy <= 1; dy++) {
        const long row = ((long)(z + dz) * pny + (yy + dy)) * pnx + x;
        any |= cs[row + 2] - cs[row - 1];                       // points in cells x-1 .. x+1 of that row
      }
  }
  occ[D.cs_off[y] + c] = any ? 1 : 0;
}


// Grow-only scratch of the grid build, one per device, handed out under a mutex (er_cloud_create may be called from
// several host threads; builds on one device then take turns).
struct GridScratch {
  std::mutex mu;
  int device = -1;
  unsigned *key[4] = {nullptr, nullptr, nullptr, nullptr}, *idx[4] = {nullptr, nullptr, nullptr, nullptr};   // two sets: [2 lane], [2 lane + 1]
  void* cub[2] = {nullptr, nullptr};
  int *bounds = nullptr, *h_bounds = nullptr;                   // 8 ints per cloud of a batch: device and its page-locked mirror (+ the initial pattern)
  size_t n_cap = 0, cub_cap = 0, bounds_cap = 0;
  hipStream_t up[2] = {nullptr, nullptr};                       // uploads: coordinates on one, normals on the other (two DMA engines keep the link busy)
  hipStream_t cs2[2] = {nullptr, nullptr};                      // grid kernels: consecutive chunks alternate between two lanes
  hipStream_t bs = nullptr;                                     // the bounding boxes of EVERY chunk (stage A): waits for uploads never sit in front of a grid
  hipEvent_t lane_ev = nullptr;
  std::vector<hipEvent_t> ev;                                   // three per chunk of a batch: uploads done (two streams), bounds back
};
GridScratch& grid_scratch(int device) {
  static GridScratch* tab = new GridScratch[64];             // intentionally leaked (see WsPool)
  return tab[device & 63];
}

// One device allocation shared by the clouds of a chunk (freed with the last of them).
struct CloudSlab {
  void* p = nullptr;
  std::atomic<int> refs{0};
};
void slab_release(CloudSlab* sl) {
  if (sl && sl->refs.fetch_sub(1) == 1) {
    if (sl->p) (void)hipFree(sl->p);
    delete sl;
  }
}

}  // namespace

struct er_cloud_s {
  int device = 0, n = 0;
  float *xyz = nullptr, *nrm = nullptr;
  float4* sorted = nullptr;
  float4* xn = nullptr;         // [2n] file order: {x, y, z, 0}, {nx, ny, nz, 0} -- ONE 32-byte gather per matched point in k_icp_iter
  int* cell_start = nullptr;
  Grid grid{};
  float radius_cap = 0.f;       // largest search radius the grid supports
  float nmax = 1.f;             // largest finite |normal component| (k_chunk_bounds): bounds the ICP sums (PairDev::fx_scale)
  CloudSlab *pts_slab = nullptr, *cell_slab = nullptr;   // the chunk's allocations these pointers live in
};

namespace {
Grid grid_of(const er_cloud_s* c) { return c->grid; }

// ---- group workspaces ----------------------------------------------------------------------------
// Everything a GROUP of pairs needs while it is being processed (clouds are immutable and shared): one compute stream and one
// copy stream, the per-pair descriptors and ICP states (device + pinned host mirrors), small per-pair result arrays, and slabs of
// per-source-point scratch cut into one slice per pair.  Groups live in a per-device pool: an API call borrows one (calls from
// several host threads -- the reference's "#pragma omp parallel for" -- borrow one each and overlap on the GPU) and returns it.
// The pool is never freed behind the HIP runtime's back (er_icp_release_workspaces does it).
struct Group {
  int device = 0;
  hipStream_t stream = nullptr, copy_stream = nullptr, copy_stream2 = nullptr;   // copy streams: list copies alternate between them ...
  hipStream_t copy_more[2] = {nullptr, nullptr};    // ... and two more (round 6; created on first use): which SDMA engine a stream's copies run on is the
                                                    // runtime's choice, and two streams that share one engine move 83 MB of lists in 4.4 instead of 1.6 ms
  hipEvent_t ev = nullptr, ev2 = nullptr;
  std::vector<hipEvent_t> sub_ev, sub_ev2;  // per sub-group of er_find_correspondence_batch (grown on demand): done / scanned
  int cap_pairs = 0;
  size_t cap_points = 0, cap_blocks = 0, cap_parts = 0;
  // device
  PairDev* d_pairs = nullptr;
  IcpDev* d_state = nullptr;
  int *d_active = nullptr, *d_counts = nullptr, *d_totals = nullptr;
  double *d_info = nullptr, *d_fit = nullptr;      // kAcc per pair; 2 per pair
  float* X = nullptr;
  int *match = nullptr, *pairs = nullptr, *block_count = nullptr, *block_offset = nullptr;
  double* partial = nullptr;
  // pinned host mirrors
  PairDev* h_pairs = nullptr;
  IcpDev* h_state = nullptr;
  int *h_active = nullptr, *h_counts = nullptr, *h_totals = nullptr;
  double *h_info = nullptr, *h_fit = nullptr;
  int* stage = nullptr;         // pinned: pair lists on their way to pageable caller memory (lazy)
  size_t stage_cap = 0;
};

void group_free_slabs(Group* g) {
  void* ptrs[] = {g->X, g->match, g->pairs, g->block_count, g->block_offset, g->partial};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  g->X = nullptr;
  g->match = g->pairs = g->block_count = g->block_offset = nullptr;
  g->partial = nullptr;
  g->cap_points = g->cap_blocks = g->cap_parts = 0;
}

void group_free_pairs(Group* g) {
  void* dev[] = {g->d_pairs, g->d_state, g->d_active, g->d_counts, g->d_totals, g->d_info, g->d_fit};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  void* host[] = {g->h_pairs, g->h_state, g->h_active, g->h_counts, g->h_totals, g->h_info, g->h_fit};
  for (void* p : host)
    if (p) (void)hipHostFree(p);
  g->d_pairs = nullptr; g->d_state = nullptr; g->d_active = g->d_counts = g->d_totals = nullptr; g->d_info = g->d_fit = nullptr;
  g->h_pairs = nullptr; g->h_state = nullptr; g->h_active = g->h_counts = g->h_totals = nullptr; g->h_info = g->h_fit = nullptr;
  g->cap_pairs = 0;
}

void group_destroy(Group* g) {
  if (!g) return;
  (void)hipSetDevice(g->device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  if (g->copy_stream) (void)hipStreamSynchronize(g->copy_stream);
  if (g->copy_stream2) (void)hipStreamSynchronize(g->copy_stream2);
  for (hipStream_t x : g->copy_more)
    if (x) {
      (void)hipStreamSynchronize(x);
      (void)hipStreamDestroy(x);
    }
  group_free_slabs(g);
  group_free_pairs(g);
  if (g->stage) (void)hipHostFree(g->stage);
  if (g->ev) (void)hipEventDestroy(g->ev);
  if (g->ev2) (void)hipEventDestroy(g->ev2);
  for (hipEvent_t e : g->sub_ev)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : g->sub_ev2)
    if (e) (void)hipEventDestroy(e);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  if (g->copy_stream) (void)hipStreamDestroy(g->copy_stream);
  if (g->copy_stream2) (void)hipStreamDestroy(g->copy_stream2);
  delete g;
}

int nblocks_of(int n) { return (std::max(n, 1) + kBlock - 1) / kBlock; }
#ifndef ER_ICP_FIXED_PTS
#define ER_ICP_FIXED_PTS 0
#endif
#ifndef ER_PRECHECK_WGS
#define ER_PRECHECK_WGS 16384    // workgroups of one k_count_inliers launch (each strides over its pair's slices; 8192: +3 % time, 32768: the same)
#endif
// Slices of 256 points per workgroup of k_icp_iter when `blocks` slices are to be searched in one launch (the 29 cross-lane sums are paid
// once per workgroup, so more slices per workgroup are cheaper as long as the launch still fills the chip).
int icp_pts(long blocks) {
  if (ER_ICP_FIXED_PTS) return ER_ICP_FIXED_PTS;
  return blocks >= 1024 ? kIcpPtsMax : 1;
}
int nparts_of(int n, int pts) { return (std::max(n, 1) + kBlock * pts - 1) / (kBlock * pts); }

// Room for `pairs` descriptors and, when per-point scratch is needed, for `points` source points in `blocks` / `parts` blocks.
int group_reserve(Group* g, int pairs, size_t points, size_t blocks, size_t parts) {
  if (pairs > g->cap_pairs) {
    ER_HIP_TRY(hipStreamSynchronize(g->stream));
    group_free_pairs(g);
    const size_t c = (size_t)std::max(pairs, 16);
    ER_HIP_TRY(hipMalloc((void**)&g->d_pairs, c * sizeof(PairDev)));
    ER_HIP_TRY(hipMalloc((void**)&g->d_state, c * sizeof(IcpDev)));
    ER_HIP_TRY(hipMalloc((void**)&g->d_active, c * sizeof(int)));
    ER_HIP_TRY(hipMalloc((void**)&g->d_counts, c * sizeof(int)));
    ER_HIP_TRY(hipMalloc((void**)&g->d_totals, c * sizeof(int)));
    ER_HIP_TRY(hipMalloc((void**)&g->d_info, c * kAcc * sizeof(double)));
    ER_HIP_TRY(hipMalloc((void**)&g->d_fit, c * 2 * sizeof(double)));
    ER_HIP_TRY(hipHostMalloc((void**)&g->h_pairs, c * sizeof(PairDev), hipHostMallocDefault));
    ER_HIP_TRY(hipHostMalloc((void**)&g->h_state, c * sizeof(IcpDev), hipHostMallocDefault));
    ER_HIP_TRY(hipHostMalloc((void**)&g->h_active, c * sizeof(int), hipHostMallocDefault));
    ER_HIP_TRY(hipHostMalloc((void**)&g->h_counts, c * sizeof(int), hipHostMallocDefault));
    ER_HIP_TRY(hipHostMalloc((void**)&g->h_totals, c * sizeof(int), hipHostMallocDefault));
    ER_HIP_TRY(hipHostMalloc((void**)&g->h_info, c * kAcc * sizeof(double), hipHostMallocDefault));
    ER_HIP_TRY(hipHostMalloc((void**)&g->h_fit, c * 2 * sizeof(double), hipHostMallocDefault));
    g->cap_pairs = (int)c;
  }
  if (points > g->cap_points || blocks > g->cap_blocks || parts > g->cap_parts) {
    ER_HIP_TRY(hipStreamSynchronize(g->stream));
    if (g->copy_stream) ER_HIP_TRY(hipStreamSynchronize(g->copy_stream));
    if (g->copy_stream2) ER_HIP_TRY(hipStreamSynchronize(g->copy_stream2));
    for (hipStream_t x : g->copy_more)
      if (x) ER_HIP_TRY(hipStreamSynchronize(x));
    const size_t cp = std::max(points + points / 8, g->cap_points), cb = std::max(blocks + blocks / 8, g->cap_blocks),
                 cq = std::max(parts + parts / 8, g->cap_parts);
    group_free_slabs(g);
    ER_HIP_TRY(hipMalloc((void**)&g->X, std::max<size_t>(cp, 1) * 3 * sizeof(float)));
    ER_HIP_TRY(hipMalloc((void**)&g->match, std::max<size_t>(cp, 1) * sizeof(int)));
    ER_HIP_TRY(hipMalloc((void**)&g->pairs, std::max<size_t>(cp, 1) * 2 * sizeof(int)));
    ER_HIP_TRY(hipMalloc((void**)&g->block_count, std::max<size_t>(cb, 1) * sizeof(int)));
    ER_HIP_TRY(hipMalloc((void**)&g->block_offset, std::max<size_t>(cb, 1) * sizeof(int)));
    ER_HIP_TRY(hipMalloc((void**)&g->partial, std::max<size_t>(cq, 1) * 64 * sizeof(double)));   // 32 x 16 bytes per part (k_icp_iter's 128-bit sums)
    g->cap_points = cp;
    g->cap_blocks = cb;
    g->cap_parts = cq;
  }
  return 0;
}

struct GroupPool {
  std::mutex mu;
  std::vector<Group*> idle;
};
GroupPool& pool() {
  static GroupPool* p = new GroupPool();      // intentionally leaked: must outlive every static destructor
  return *p;
}

Group* group_acquire(int device) {
  Group* g = nullptr;
  {
    std::lock_guard<std::mutex> lock(pool().mu);
    // the idle workspace with the LARGEST scratch (ties: the one released last).  Until round 6 this took the first one: after a call that had used
    // several workspaces at once (er_registration_batch's shares, eight host threads with a pair each) a single-threaded caller rotated through all
    // of them, and every one sized for a share had to free and re-allocate its slabs for the full list -- 6.4 against 9.1 ms per 50-pair list,
    // alternating, in bench.py's kinfu-like leg (BENCH_r05: 5.5 k pairs/s where the kernels deliver 7.9 k).
    auto& v = pool().idle;
    long best = -1;
    for (size_t i = 0; i < v.size(); i++)
      if (v[i]->device == device && (best < 0 || v[i]->cap_points >= v[(size_t)best]->cap_points)) best = (long)i;
    if (best >= 0) {
      g = v[(size_t)best];
      v.erase(v.begin() + best);
    }
  }
  if (hipSetDevice(device) != hipSuccess) {
    er::fail("hipSetDevice(%d) failed", device);
    return nullptr;
  }
  if (!g) {
    g = new Group();
    g->device = device;
    if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&g->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&g->copy_stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&g->ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->ev2, hipEventDisableTiming) != hipSuccess) {
      er::fail("ICP workspace allocation failed: %s", hipGetErrorString(hipGetLastError()));
      group_destroy(g);
      return nullptr;
    }
  }
  return g;
}

struct GroupLease {             // RAII: the group of one API call
  Group* g = nullptr;
  ~GroupLease() {
    if (!g) return;
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    if (g->copy_stream) (void)hipStreamSynchronize(g->copy_stream);
    if (g->copy_stream2) (void)hipStreamSynchronize(g->copy_stream2);
    for (hipStream_t x : g->copy_more)
      if (x) (void)hipStreamSynchronize(x);
    std::lock_guard<std::mutex> lock(pool().mu);
    pool().idle.push_back(g);
  }
  int acquire(int device) {
    g = group_acquire(device);
    return g ? 0 : 1;
  }
};

// Pairs per group: bounded by the per-point scratch (32 bytes per source point and pair).  ER_ICP_GROUP overrides (tests use small groups).
static int group_cfg() { const char* e = getenv("ER_ICP_GROUP"); int v = e ? atoi(e) : 64; return v < 1 ? 1 : (v > 1024 ? 1024 : v); }

int check_pair(er_cloud_t src, er_cloud_t tgt, double radius, const char* who) {
  if (!src || !tgt) return er::fail("%s: NULL cloud", who);
  if (src->device != tgt->device) return er::fail("%s: source and target live on different devices", who);
  if (!(radius > 0.0) || radius > (double)tgt->radius_cap * (1.0 + 1e-6))
    return er::fail("%s: search radius %g exceeds the target's grid cell %g (er_cloud_create grid_cell)", who, radius, (double)tgt->radius_cap);
  return 0;
}

int batch_prologue(int n, const er_cloud_t* src, const er_cloud_t* tgt, double radius, const char* who, int* device) {
  if (n < 0 || (n > 0 && (!src || !tgt))) return er::fail("%s: bad arguments", who);
  *device = n > 0 && src[0] ? src[0]->device : 0;
  for (int i = 0; i < n; i++) {
    if (check_pair(src[i], tgt[i], radius, who)) return 1;
    if (src[i]->device != *device) return er::fail("%s: all pairs of one batch must live on one device", who);
  }
  return 0;
}

// Power-of-two scales of k_icp_iter's 128-bit fixed-point partial sums: scale_k = 2^(120 - ilogb(T_k)) with T_k a CERTAIN bound of |sum k| -- n_source terms,
// each bounded through  |s| <= M (a matched source point lies within the radius of a target point, i.e. within the target's box + radius),
// |normal component| <= nm (k_chunk_bounds), |d - s| <= r, |e| = |n . (d - s)| <= 3 nm (r + 1e-4 M) (the float32 expression cancels terms of size nm M) --
// so the 128-bit total cannot overflow (|sum| * scale < 2^121) whatever the points are.
void icp_fixed_point_scales(int n_src, const er_cloud_s* t, double* scale, double* inv) {
  double M = 0.0;
  const double r = 1.01 * (double)t->radius_cap;
  for (int a = 0; a < 3; a++)
    M = std::max(M, std::max(std::fabs((double)t->grid.org[a]), std::fabs((double)t->grid.org[a] + (double)(t->grid.dim[a] + 1) * (double)t->grid.cell)) + r);
  const double nm = std::max((double)t->nmax, 1e-30), n = 1.1 * (double)std::max(n_src, 1), e = 3.0 * nm * (r + 1e-4 * M);
  const double rot = 2.0 * nm * M;
  // classes: rot x rot, rot x normal, normal x normal, rot x e, normal x e, squared distance, count
  const double T[7] = {n * rot * rot, n * rot * nm, n * nm * nm, n * rot * e, n * nm * e, n * r * r, n};
  static const int cls[29] = {0, 0, 0, 1, 1, 1, 0, 0, 1, 1, 1, 0, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6};
  for (int k = 0; k < 32; k++) {
    double sc = 0.0;
    if (k < 29) {
      const double b = T[cls[k]];
      sc = (std::isfinite(b) && b > 0.0) ? std::ldexp(1.0, 120 - std::ilogb(b)) : 0.0;
    }
    scale[k] = sc;
    inv[k] = sc > 0.0 ? 1.0 / sc : 0.0;
  }
}

// Fills the descriptors of pairs [i0, i0 + m) of the caller's lists into slots 0..m-1 and cuts the scratch slabs (scratch = false:
// the pre-check needs none).  T16: one row-major float64 4x4 per pair, or NULL.
int group_describe(Group* g, int i0, int m, const er_cloud_t* src, const er_cloud_t* tgt, const double* T16, bool scratch, int* const* pairs_direct = nullptr) {
  size_t points = 0, blocks = 0, parts = 0;
  // (the per-workgroup partial sums of k_icp_iter: room for one slice of 256 points per workgroup, the finest split icp_pts below can pick)
  const int pts = 1;
  for (int q = 0; q < m; q++) {
    const int n = src[i0 + q]->n;
    points += (size_t)((n + 3) & ~3);                          // slices stay 16-byte aligned
    blocks += (size_t)nblocks_of(n);
    parts += (size_t)nparts_of(n, pts);
  }
  if (group_reserve(g, m, scratch ? points : 0, scratch ? blocks : 0, scratch ? parts : 0)) return 1;
  size_t op = 0, ob = 0, oq = 0;
  for (int q = 0; q < m; q++) {
    const er_cloud_s* s = src[i0 + q];
    const er_cloud_s* t = tgt[i0 + q];
    PairDev& P = g->h_pairs[q];
    memset(&P, 0, sizeof P);
    P.src_sorted = s->sorted; P.src_xyz = s->xyz; P.src_nrm = s->nrm;
    P.tgt_xyz = t->xyz; P.tgt_nrm = t->nrm; P.tgt_xn = t->xn;
    P.g = grid_of(t);
    if (T16)
      for (int c = 0; c < 12; c++) P.T.m[c] = T16[(size_t)(i0 + q) * 16 + c];
    P.n = s->n; P.nb = nblocks_of(s->n); P.nbi = nparts_of(s->n, pts); P.pts = pts;
    icp_fixed_point_scales(s->n, t, P.fx_scale, P.fx_inv);
    if (scratch) {
      P.X = g->X + 3 * op; P.match = g->match + op; P.pairs = g->pairs + 2 * op;
      P.block_count = g->block_count + ob; P.block_offset = g->block_offset + ob;
      P.partial = g->partial + 64 * oq;
      if (pairs_direct && pairs_direct[i0 + q]) P.pairs = pairs_direct[i0 + q];   // k_compact writes the list where the caller wants it (see er_find_correspondence_batch)
      op += (size_t)((s->n + 3) & ~3); ob += (size_t)P.nb; oq += (size_t)P.nbi;
    }
  }
  ER_HIP_TRY(hipMemcpyAsync(g->d_pairs, g->h_pairs, (size_t)m * sizeof(PairDev), hipMemcpyHostToDevice, g->stream));
  return 0;
}

int max_points(int i0, int m, const er_cloud_t* src) {
  int mx = 0;
  for (int q = 0; q < m; q++) mx = std::max(mx, src[i0 + q]->n);
  return mx;
}

// Lists are written in place by k_compact when the GPU can address the destination: device memory (no device-to-device copy) and page-locked host
// memory (no hipMemcpyAsync).  For a host destination that is 2.1-2.5 ms per 50-pair list whatever the state of the process: the copies it replaces
// took 1.6-1.9 ms when their two streams ran on separate SDMA engines and 4.3-4.7 ms when they shared one -- the runtime's choice, whole processes
// long (profiles/r06h_realistic_probe.txt, r06l_fc_modes.txt).  ER_ICP_DIRECT_LISTS=0 copies everything, =d only device destinations are written in place.
static int direct_lists() {
  static const int mode = [] { const char* e = getenv("ER_ICP_DIRECT_LISTS"); return e ? (e[0] == '0' ? 0 : (e[0] == 'd' ? 1 : 2)) : 2; }();
  return mode;                                                 // 0: never, 1: device destinations, 2: device and page-locked host destinations
}
// Set by er_registration_batch's share workers: several shares are in flight, each on a thread and workspace of its own, so a share's list copies run
// behind the other shares' kernels -- and six shares' PCIe-bound compaction kernels at once are slower than their copies (fused list 6.05 against 4.28 ms,
// profiles/r06q_fused_modes.txt).  A single caller has nothing to hide a copy behind: its lists are written in place.
static thread_local bool tl_lists_by_copy = false;

// Copy streams the list copies of one call are dealt to when lists ARE copied (pageable or too small destinations): ER_ICP_COPY_STREAMS = 1 .. 4,
// default 2 (four streams: 2.5-3.0 ms whatever the engines do, two: 1.6 or 4.4 ms; profiles/r06j_fc_modes.txt).
static int list_copy_streams() {
  static const int n = [] { const char* e = getenv("ER_ICP_COPY_STREAMS"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > 4 ? 4 : v); }();
  return n;
}

// Where a caller's list buffer lives: 0 = pageable host memory (staged through the group's page-locked block), 1 = page-locked host memory (the copy
// goes straight there), 2 = DEVICE memory of `device` (round 5: the list stays in HBM -- a device-to-device copy of exactly the list, no PCIe; what a
// consumer on the same GPU wants, er_fopt_set_correspondences_dev), -1 = device memory of another GPU (refused).
int buffer_kind(const void* p, int device, void** device_view = nullptr) {
  hipPointerAttribute_t at;
  if (device_view) *device_view = nullptr;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();                                 // plain malloc memory: "invalid value", not an error for us
    return 0;
  }
  if (at.type == hipMemoryTypeHost) {
    if (device_view) *device_view = at.devicePointer;       // page-locked host memory is mapped into the device's address space
    return 1;
  }
  if (at.type == hipMemoryTypeDevice) {
    if (device_view && at.device == device) *device_view = const_cast<void*>(p);
    return at.device == device ? 2 : -1;
  }
  return 0;
}
bool is_pinned_host(const void* p) { return buffer_kind(p, -1) == 1; }

// sum A^T A with A = [I | B], B = [[0, 2sz, -2sy], [-2sz, 0, 2sx], [2sy, -2sx, 0]]  (CorresApp.cpp:198-203,
// RansacCurvature.h:717-721) from its ten distinct terms (k_count_blocks).
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

// Iterations enqueued per host visit (the first visit; later ones twice as many).  PCL's loop runs 3 iterations on most fragment pairs of the
// pipeline (the third one meets the stop rule) and 4 on a few: one chunk usually ends the job; workgroups of a chunk that come after a pair's
// stop decision return at once.
constexpr int kIcpChunk = 4;
static int icp_chunk() { const char* e = getenv("ER_ICP_CHUNK"); const int v = e ? atoi(e) : kIcpChunk; return v < 1 ? 1 : (v > 20 ? 20 : v); }   // (A/B switch)

}  // namespace

extern "C" {

// Clouds of a LIST of fragments (BuildCorrespondence's LoadData loop, CorresApp.cpp:82-110, builds them one after the other), in chunks of up to
// kCloudChunk clouds that share two device allocations and ONE set of grid launches:
//   stage A, for every chunk up front: the chunk's allocation, its uploads -- coordinates and normals on two copy streams, truly asynchronous
//            when the caller's arrays are page-locked (er_host_alloc) -- and its bounding boxes on a compute lane as soon as the uploads are in;
//   stage B, chunk by chunk: ONE host wait for the boxes (the host sizes the grids from them), then cell ids + histogram + interleave, one
//            radix sort, one prefix sum, one gather for the whole chunk, while the later chunks are still uploading.
// The list is PCIe-bound; before round 4's last step it was bound by the host's launch rate (23 launches per cloud).
int er_cloud_create_batch(int n_clouds, const float* const* xyz_host, const float* const* normal_host, const int* counts, float grid_cell, int device,
                          er_cloud_t* out) {
  if (n_clouds < 0 || (n_clouds > 0 && (!xyz_host || !normal_host || !counts || !out))) return er::fail("er_cloud_create: bad arguments");
  for (int i = 0; i < n_clouds; i++) out[i] = nullptr;
  for (int i = 0; i < n_clouds; i++)
    if (counts[i] < 0 || (counts[i] > 0 && (!xyz_host[i] || !normal_host[i]))) return er::fail("er_cloud_create: bad arguments (cloud %d)", i);
  // the search kernels address candidates, matched-target records and the ICP loop's X by UNSIGNED 32-bit byte offsets (s * 16, i * 32, k * 12: struct
  // PairDev's hot accesses): a cloud must stay below 2^27 points, or those offsets wrap and the kernels read other memory
  for (int i = 0; i < n_clouds; i++)
    if (counts[i] >= (1 << 27)) return er::fail("er_cloud_create: cloud %d has %d points; the limit is 2^27 - 1 (32-bit byte offsets in the search kernels)", i, counts[i]);
  if (!(grid_cell > 0.f)) return er::fail("er_cloud_create: grid_cell must be positive");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return er::fail("er_cloud_create: no HIP device available (liber_hip has no CPU fallback)");
  if (device < 0 || device >= ndev) return er::fail("er_cloud_create: device %d out of range [0,%d)", device, ndev);
  if (n_clouds == 0) return 0;
  ER_HIP_TRY(hipSetDevice(device));
  GridScratch& gs = grid_scratch(device);
  std::lock_guard<std::mutex> lock(gs.mu);
  auto sync_all = [&]() -> hipError_t {
    hipError_t e = hipSuccess;
    for (hipStream_t st : {gs.up[0], gs.up[1], gs.cs2[0], gs.cs2[1], gs.bs})
      if (st) {
        const hipError_t e1 = hipStreamSynchronize(st);
        if (e == hipSuccess) e = e1;
      }
    return e;
  };
  auto undo = [&]() {
    (void)sync_all();
    for (int i = 0; i < n_clouds; i++) {
      if (out[i]) er_cloud_destroy(out[i]);
      out[i] = nullptr;
    }
  };
#define ER_CTRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      er::fail("er_cloud_create: %s failed: %s", #expr, hipGetErrorString(e_));               \
      undo();                                                                                 \
      return 1;                                                                               \
    }                                                                                         \
  } while (0)
  for (int q = 0; q < 2; q++) {
    if (!gs.up[q]) ER_CTRY(hipStreamCreateWithFlags(&gs.up[q], hipStreamNonBlocking));
    if (!gs.cs2[q]) ER_CTRY(hipStreamCreateWithFlags(&gs.cs2[q], hipStreamNonBlocking));
    if (!gs.bs) ER_CTRY(hipStreamCreateWithFlags(&gs.bs, hipStreamNonBlocking));
  }
  if (!gs.lane_ev) ER_CTRY(hipEventCreateWithFlags(&gs.lane_ev, hipEventDisableTiming));
  // ---- the chunks: consecutive clouds, at most kCloudChunk of them and kChunkPoints points (the first cloud of a chunk always fits) ----
  constexpr long kChunkPoints = 16L << 20;
  struct Chunk {
    int i0 = 0, i1 = 0;
    long total = 0;                     // points
    CloudSlab *pts = nullptr, *cells = nullptr;
    float4* sorted = nullptr;           // the chunk's sorted array (the clouds' pieces are consecutive)
    ChunkDesc D{};
  };
  std::vector<Chunk> chunks;
  {
    // a chunk takes half of what is left (at most kCloudChunk): the list ends in small chunks, and what remains to be done after the
    // last upload -- the last chunk's grids -- is short (25 fragments: 8, 8, 5, 2, 1, 1)
    for (int i = 0; i < n_clouds;) {
      Chunk c;
      c.i0 = i;
      const int per = std::min(kCloudChunk, std::max(1, (n_clouds - i + 1) / 2));
      while (i < n_clouds && i - c.i0 < per && (i == c.i0 || c.total + counts[i] <= kChunkPoints)) c.total += counts[i++];
      c.i1 = i;
      chunks.push_back(c);
    }
  }
  const int n_chunks = (int)chunks.size();
  while ((int)gs.ev.size() < 3 * n_chunks) {
    hipEvent_t e = nullptr;
    ER_CTRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    gs.ev.push_back(e);
  }
  if (gs.bounds_cap < (size_t)n_clouds) {
    ER_CTRY(sync_all());
    if (gs.bounds) (void)hipFree(gs.bounds);
    if (gs.h_bounds) (void)hipHostFree(gs.h_bounds);
    gs.bounds = gs.h_bounds = nullptr;
    gs.bounds_cap = 0;
    const size_t cap = (size_t)n_clouds + 16;
    ER_CTRY(hipMalloc((void**)&gs.bounds, cap * 8 * sizeof(int)));
    ER_CTRY(hipHostMalloc((void**)&gs.h_bounds, cap * 16 * sizeof(int), hipHostMallocDefault));   // [cap][8] results, then [cap][8] initial pattern
    gs.bounds_cap = cap;
  }
  long t_max = 0;
  for (const Chunk& c : chunks) t_max = std::max(t_max, c.total);
  if (t_max >= (1L << 31) - kBlock) return er::fail("er_cloud_create: a cloud of %ld points is beyond the 32-bit point index", t_max);
  if (t_max > 0) {
    if (gs.n_cap < (size_t)t_max) {
      ER_CTRY(sync_all());
      for (int q = 0; q < 4; q++) {
        if (gs.key[q]) (void)hipFree(gs.key[q]);
        if (gs.idx[q]) (void)hipFree(gs.idx[q]);
        gs.key[q] = gs.idx[q] = nullptr;
      }
      gs.n_cap = 0;
      const size_t cap = (size_t)t_max + (size_t)t_max / 8;
      for (int q = 0; q < 4; q++) {
        ER_CTRY(hipMalloc((void**)&gs.key[q], cap * sizeof(unsigned)));
        ER_CTRY(hipMalloc((void**)&gs.idx[q], cap * sizeof(unsigned)));
      }
      gs.n_cap = cap;
    }
    // temporary storage of the sort / scan for the largest chunk and the largest grids (2^25 cells and 3 cloud bits: 28 key bits) this call can meet
    size_t need_sort = 0, need_scan = 0;
    ER_CTRY(hipcub::DeviceRadixSort::SortPairs(nullptr, need_sort, gs.key[0], gs.key[1], gs.idx[0], gs.idx[1], (int)t_max, 0, 28, gs.cs2[0]));
    ER_CTRY(hipcub::DeviceScan::InclusiveSum(nullptr, need_scan, (int*)nullptr, (int*)nullptr, kCloudChunk * ((1 << 25) + 1), gs.cs2[0]));
    const size_t need = std::max(need_sort, need_scan);
    if (gs.cub_cap < need) {
      ER_CTRY(sync_all());
      for (int q = 0; q < 2; q++) {
        if (gs.cub[q]) (void)hipFree(gs.cub[q]);
        gs.cub[q] = nullptr;
      }
      gs.cub_cap = 0;
      for (int q = 0; q < 2; q++) ER_CTRY(hipMalloc(&gs.cub[q], need + need / 4));
      gs.cub_cap = need + need / 4;
    }
  }
  int* h_got = gs.h_bounds;
  int* h_init = gs.h_bounds + gs.bounds_cap * 8;
  for (int i = 0; i < n_clouds; i++) {
    const int init[8] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN, 0, 0};
    memcpy(h_init + (size_t)i * 8, init, sizeof init);
  }
  ER_CTRY(hipMemcpyAsync(gs.bounds, h_init, (size_t)n_clouds * 8 * sizeof(int), hipMemcpyHostToDevice, gs.cs2[0]));
  ER_CTRY(hipEventRecord(gs.lane_ev, gs.cs2[0]));
  ER_CTRY(hipStreamWaitEvent(gs.cs2[1], gs.lane_ev, 0));
  ER_CTRY(hipStreamWaitEvent(gs.bs, gs.lane_ev, 0));            // (the boxes' initial pattern is in place before the first bounds kernel)
  // ---- stage A of a chunk: allocation, uploads (two copy streams), bounding boxes (the chunk's compute lane) ----
  auto stage_a = [&](int ch) -> int {
    Chunk& C = chunks[(size_t)ch];
    const int m = C.i1 - C.i0;
    const size_t N = (size_t)C.total;
    // one allocation: [sorted N float4 + kSentinel entries of +inf behind every cloud (scan_range) | xn 2N float4 (32-byte aligned records) | per cloud: xyz 3n,
    // normals 3n floats]
    const size_t xn_base = (N + (size_t)kSentinel * (size_t)m + 1) & ~(size_t)1;
    const size_t f_base = (xn_base + 2 * N) * 4;
    const size_t bytes = std::max((f_base + 6 * N) * sizeof(float), (size_t)256);
    C.pts = new CloudSlab();
    hipError_t e = hipMalloc(&C.pts->p, bytes);
    if (e != hipSuccess) {
      delete C.pts;
      C.pts = nullptr;
      return er::fail("er_cloud_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    C.pts->refs = m;
    C.sorted = static_cast<float4*>(C.pts->p);
    float* fbase = static_cast<float*>(C.pts->p) + f_base;
    C.D.m = m;
    long off = 0;
    for (int k = 0; k < m; k++) {                                 // the cloud objects first: from here on undo() releases the allocation through them
      const int i = C.i0 + k, n = counts[i];
      er_cloud_t c = new er_cloud_s();
      out[i] = c;
      c->device = device;
      c->n = n;
      c->radius_cap = grid_cell;
      c->pts_slab = C.pts;
      c->sorted = C.sorted + off + (long)kSentinel * k;
      c->xn = C.sorted + xn_base + 2 * off;
      c->xyz = fbase + 6 * off;
      c->nrm = c->xyz + 3 * (size_t)n;
      C.D.n[k] = n;
      C.D.pt_off[k] = (int)off;
      C.D.xyz[k] = c->xyz;
      C.D.nrm[k] = c->nrm;
      C.D.xn[k] = c->xn;
      off += n;
    }
    C.D.pt_off[m] = (int)off;
    for (int k = 0; k < m; k++) {
      const int i = C.i0 + k, n = counts[i];
      if (n > 0) {
        ER_HIP_TRY(hipMemcpyAsync(out[i]->xyz, xyz_host[i], (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, gs.up[0]));
        ER_HIP_TRY(hipMemcpyAsync(out[i]->nrm, normal_host[i], (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, gs.up[1]));
      }
    }
    // Stage A runs on a stream of its own (round 5): on the two grid lanes the upload waits of chunk 2, 4, ... were queued IN FRONT of
    // chunk 0's grid (in-order streams; every stage A is enqueued before the first stage B), so "built while the later chunks are still uploading" mostly
    // did not happen.  Stage B of a chunk starts after the HOST has seen the chunk's boxes (ev[3 ch + 2]), hence after its uploads.
    hipStream_t L = gs.bs;
    for (int q = 0; q < 2; q++) {
      ER_HIP_TRY(hipEventRecord(gs.ev[(size_t)(3 * ch + q)], gs.up[q]));
      ER_HIP_TRY(hipStreamWaitEvent(L, gs.ev[(size_t)(3 * ch + q)], 0));
    }
    int n_big = 0;
    for (int k = 0; k < m; k++) n_big = std::max(n_big, C.D.n[k]);
    if (n_big > 0)
      hipLaunchKernelGGL(k_chunk_bounds, dim3(std::min(nblocks_of(n_big), 128), m), dim3(kBlock), 0, L, C.D, gs.bounds + (size_t)C.i0 * 8);
    ER_HIP_TRY(hipGetLastError());
    ER_HIP_TRY(hipMemcpyAsync(h_got + (size_t)C.i0 * 8, gs.bounds + (size_t)C.i0 * 8, (size_t)m * 8 * sizeof(int), hipMemcpyDeviceToHost, L));
    ER_HIP_TRY(hipEventRecord(gs.ev[(size_t)(3 * ch + 2)], L));
    return 0;
  };
  // ---- stage B: the grids of the chunk (its boxes are back) ----
  auto stage_b = [&](int ch) -> int {
    Chunk& C = chunks[(size_t)ch];
    const int m = C.i1 - C.i0;
    ER_HIP_TRY(hipEventSynchronize(gs.ev[(size_t)(3 * ch + 2)]));
    long cs_total = 0;
    int max_cells = 1;
    for (int k = 0; k < m; k++) {
      er_cloud_t c = out[C.i0 + k];
      const int n = c->n;
      const int* got = h_got + (size_t)(C.i0 + k) * 8;
      float cell = grid_cell * 1.001f;                            // strictly larger than any admissible radius
      float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
      int dim[3] = {1, 1, 1};
      if (n > 0) {
        if (got[7] > 0) memcpy(&c->nmax, &got[7], sizeof(float));
        if (got[6]) return er::fail("er_cloud_create: non-finite coordinates");
        for (int a = 0; a < 3; a++) {
          lo[a] = ordered_float(got[a]);
          hi[a] = ordered_float(got[3 + a]);
        }
      }
      constexpr int kRing = 4;                                    // two rings of empty cells per axis (struct Grid)
      for (;;) {
        long total = 1;
        for (int a = 0; a < 3; a++) {
          dim[a] = (int)std::floor((hi[a] - lo[a]) / cell) + 1;
          total *= dim[a] + kRing;
        }
        if (total <= (1L << 25)) break;
        cell *= 2.f;
      }
      const int ncell = (dim[0] + kRing) * (dim[1] + kRing) * (dim[2] + kRing);
      max_cells = std::max(max_cells, ncell);
      C.D.cs_off[k] = cs_total;
      cs_total += (long)ncell + 1;
      c->grid.cell = cell;
      for (int a = 0; a < 3; a++) {
        c->grid.org[a] = C.D.G[k].org[a] = lo[a];
        c->grid.dim[a] = C.D.G[k].dim[a] = dim[a];
      }
      C.D.G[k].cell = cell;
      c->grid.slack = grid_slack(dim, cell);
      c->grid.pnx = dim[0] + 4;
      c->grid.pny = dim[1] + 4;
    }
    C.D.cs_off[m] = cs_total;
    C.cells = new CloudSlab();
    const size_t cells_bytes = (size_t)cs_total * sizeof(int) + (size_t)cs_total;   // cell_start of the chunk's clouds, then one occupancy byte per cell
    hipError_t e = hipMalloc(&C.cells->p, cells_bytes);
    if (e != hipSuccess) {
      delete C.cells;
      C.cells = nullptr;
      return er::fail("er_cloud_create: hipMalloc(%zu) failed: %s", cells_bytes, hipGetErrorString(e));
    }
    C.cells->refs = m;
    int* cs = static_cast<int*>(C.cells->p);
    unsigned char* occ = reinterpret_cast<unsigned char*>(cs + cs_total);
    for (int k = 0; k < m; k++) {
      er_cloud_t c = out[C.i0 + k];
      c->cell_slab = C.cells;
      c->cell_start = cs + C.D.cs_off[k];
      c->grid.pts = c->sorted;
      c->grid.cell_start = c->cell_start;
      c->grid.occ = occ + C.D.cs_off[k];
    }
    int bits = 1;
    while ((1L << bits) < (long)max_cells) bits++;
    C.D.shift = bits;
    const int q = ch & 1;
    hipStream_t L = gs.cs2[q];
    unsigned *k0 = gs.key[2 * q], *k1 = gs.key[2 * q + 1], *x0 = gs.idx[2 * q], *x1 = gs.idx[2 * q + 1];
    ER_HIP_TRY(hipMemsetAsync(cs, 0, (size_t)cs_total * sizeof(int), L));
    int n_big = 0;
    for (int k = 0; k < m; k++) n_big = std::max(n_big, C.D.n[k]);
    hipLaunchKernelGGL(k_chunk_cells, dim3(nblocks_of(n_big), m), dim3(kBlock), 0, L, C.D, k0, x0, cs);
    size_t tmp = gs.cub_cap;
    if (C.total > 0) ER_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(gs.cub[q], tmp, k0, k1, x0, x1, (int)C.total, 0, bits + 3, L));
    tmp = gs.cub_cap;
    ER_HIP_TRY(hipcub::DeviceScan::InclusiveSum(gs.cub[q], tmp, cs, cs, (int)cs_total, L));
    hipLaunchKernelGGL(k_chunk_occ, dim3((unsigned)((max_cells + kBlock - 1) / kBlock), m), dim3(kBlock), 0, L, C.D, cs, occ);
    if (C.total > 0) hipLaunchKernelGGL(k_chunk_gather, dim3(nblocks_of((int)C.total)), dim3(kBlock), 0, L, C.D, k1, x1, (int)C.total, C.sorted);
    ER_HIP_TRY(hipGetLastError());
    return 0;
  };
  int rc = 0;
  for (int ch = 0; ch < n_chunks && rc == 0; ch++) rc = stage_a(ch);   // every upload is queued before the first host wait
  for (int ch = 0; ch < n_chunks && rc == 0; ch++) rc = stage_b(ch);
  if (rc == 0 && sync_all() != hipSuccess)                              // the caller's arrays and the shared scratch are free again
    rc = er::fail("er_cloud_create: %s", hipGetErrorString(hipGetLastError()));
#undef ER_CTRY
  if (rc) {
    const std::string why = er_last_error();
    undo();
    return er::fail("%s", why.c_str());
  }
  return 0;
}

int er_cloud_create(const float* xyz_host, const float* normal_host, int n, float grid_cell, int device, er_cloud_t* out) {
  if (!out) return er::fail("er_cloud_create: out is NULL");
  *out = nullptr;
  if (n < 0 || (n > 0 && (!xyz_host || !normal_host))) return er::fail("er_cloud_create: bad arguments");
  return er_cloud_create_batch(1, &xyz_host, &normal_host, &n, grid_cell, device, out);
}

int er_cloud_destroy(er_cloud_t c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  slab_release(c->pts_slab);                    // xyz, normals, sorted and xn live in the chunk's allocation, cell_start in its second one:
  slab_release(c->cell_slab);                   // both go with the last cloud of the chunk
  delete c;
  return 0;
}

int er_cloud_size(er_cloud_t c) { return c ? c->n : -1; }

int er_icp_release_workspaces(void) {
  std::vector<Group*> v;
  {
    std::lock_guard<std::mutex> lock(pool().mu);
    v.swap(pool().idle);
  }
  for (Group* g : v) group_destroy(g);
  return 0;
}

// Registration pre-check of a pair list (CorresApp.cpp:249-264): one launch per group of pairs.
int er_icp_count_inliers_batch(int n, const er_cloud_t* src, const er_cloud_t* tgt, const double* T, double max_dist, int* counts) {
  int device;
  if (n > 0 && (!T || !counts)) return er::fail("er_icp_count_inliers_batch: NULL argument");
  if (batch_prologue(n, src, tgt, max_dist, "er_icp_count_inliers", &device)) return 1;
  if (n == 0) return 0;
  GroupLease L;
  if (L.acquire(device)) return 1;
  Group* g = L.g;
  const int G = group_cfg();
  for (int i0 = 0; i0 < n; i0 += G) {
    const int m = std::min(G, n - i0);
    if (group_describe(g, i0, m, src, tgt, T, false)) return 1;
    ER_HIP_TRY(hipMemsetAsync(g->d_counts, 0, (size_t)m * sizeof(int), g->stream));
    const int mx = max_points(i0, m, src);
    if (mx > 0) {
      // a few thousand workgroups in flight fill the chip; each strides over its pair's points (one atomic per workgroup)
      const int bx = std::max(1, std::min(nblocks_of(mx), std::max(64, ER_PRECHECK_WGS / m)));
      hipLaunchKernelGGL(k_count_inliers, dim3(bx, m), dim3(kBlock), 0, g->stream, g->d_pairs, (float)max_dist, max_dist * max_dist, g->d_counts);
      ER_HIP_TRY(hipGetLastError());
    }
    ER_HIP_TRY(hipMemcpyAsync(g->h_counts, g->d_counts, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, g->stream));
    ER_HIP_TRY(hipStreamSynchronize(g->stream));
    for (int q = 0; q < m; q++) counts[i0 + q] = (src[i0 + q]->n > 0 && tgt[i0 + q]->n > 0) ? g->h_counts[q] : 0;
  }
  return 0;
}

int er_icp_count_inliers(er_cloud_t src, er_cloud_t tgt, const double T[16], double max_dist, int* count) {
  if (!T || !count) return er::fail("er_icp_count_inliers: NULL argument");
  return er_icp_count_inliers_batch(1, &src, &tgt, T, max_dist, count);
}

// icp.align of a pair list (CorresApp.cpp:295-306).  Per group: the states go up once; chunks of kIcpChunk iterations are
// enqueued for the pairs that are still running -- two launches per iteration for the whole group -- and the states come back
// once per chunk; the host only compacts the list of running pairs.
int er_icp_align_batch(int n, const er_cloud_t* src, const er_cloud_t* tgt, const float* guess, double max_dist, int max_iter,
                       double transformation_epsilon, int stop_rule, float* out, int* iterations, int* converged, double* fitness) {
  int device;
  if (n > 0 && (!guess || !out)) return er::fail("er_icp_align: NULL argument");
  if (batch_prologue(n, src, tgt, max_dist, "er_icp_align", &device)) return 1;
  if (n == 0) return 0;
  GroupLease L;
  if (L.acquire(device)) return 1;
  Group* g = L.g;
  const IcpParams prm{transformation_epsilon, max_iter, stop_rule};
  const int G = group_cfg();
  for (int i0 = 0; i0 < n; i0 += G) {
    const int m = std::min(G, n - i0);
    if (group_describe(g, i0, m, src, tgt, nullptr, true)) return 1;
    int n_active = 0;
    for (int q = 0; q < m; q++) {
      IcpDev& st = g->h_state[q];
      memset(&st, 0, sizeof st);
      const float* gq = guess + (size_t)(i0 + q) * 16;
      memcpy(st.fin, gq, sizeof st.fin);                         // final_transformation_ = guess
      bool ident = true;
      for (int i = 0; i < 16; i++) {
        st.delta[i] = st.prev_delta[i] = (i % 5 == 0) ? 1.f : 0.f;
        ident = ident && gq[i] == ((i % 5 == 0) ? 1.f : 0.f);
      }
      st.prev_mse = DBL_MAX;
      st.init_apply = ident ? 0 : 1;
      if (src[i0 + q]->n == 0 || tgt[i0 + q]->n == 0) {
        st.done = 1;                                             // fewer than 3 correspondences by construction: not converged, the guess comes back
      } else {                                                   // (max_iter <= 0 still runs ONE iteration, like PCL's do { } while loop)
        g->h_active[n_active++] = q;
      }
    }
    ER_HIP_TRY(hipMemcpyAsync(g->d_state, g->h_state, (size_t)m * sizeof(IcpDev), hipMemcpyHostToDevice, g->stream));
    int chunk_no = 0;
    while (n_active > 0) {
      // the first chunk ends the job for most pairs; the stragglers then run twice as many iterations per host visit.  A launch pair whose pairs
      // are all done costs ~12 us (every workgroup leaves at once), a host visit ~60 us plus the ramp of the next launch: chunks err on the long side
      // (round 5: 4 then 8 iterations instead of 3 then 6 -- the one pair of the bench list that needs a fourth iteration no longer costs a second visit)
      const int chunk_len = chunk_no++ == 0 ? icp_chunk() : 2 * icp_chunk();
      ER_HIP_TRY(hipMemcpyAsync(g->d_active, g->h_active, (size_t)n_active * sizeof(int), hipMemcpyHostToDevice, g->stream));
      // points per thread of k_icp_iter for THIS chunk: as many slices of 256 points per workgroup as still leave ~4 workgroups per CU in
      // flight -- 8 for a full list, 1 for the two or three stragglers of a hard list on their way to the iteration limit (round 4: the
      // split used to be fixed when the group was described, so three pairs in their 20th iteration ran as 369 workgroups of eight
      // sequential searches each on a chip that holds 2048)
      long act_blocks = 0;
      int mxb = 1;
      for (int a = 0; a < n_active; a++) {
        act_blocks += g->h_pairs[g->h_active[a]].nb;
        mxb = std::max(mxb, g->h_pairs[g->h_active[a]].nb);
      }
      const int pts = icp_pts(act_blocks);
      const int mxp = (mxb + pts - 1) / pts;
      for (int c = 0; c < chunk_len; c++) {
        hipLaunchKernelGGL(k_icp_iter, dim3(mxp, n_active), dim3(kBlock), 0, g->stream, g->d_pairs, g->d_active, g->d_state, (float)max_dist,
                           max_dist * max_dist, pts);
        hipLaunchKernelGGL(k_icp_final, dim3(n_active), dim3(kBlock), 0, g->stream, g->d_pairs, g->d_active, g->d_state, prm, pts);
      }
      ER_HIP_TRY(hipGetLastError());
      ER_HIP_TRY(hipMemcpyAsync(g->h_state, g->d_state, (size_t)m * sizeof(IcpDev), hipMemcpyDeviceToHost, g->stream));
      ER_HIP_TRY(hipStreamSynchronize(g->stream));               // (d_active is free again: the chunk has run)
      int k = 0;
      for (int a = 0; a < n_active; a++)
        if (!g->h_state[g->h_active[a]].done) g->h_active[k++] = g->h_active[a];
      n_active = k;
    }
    if (fitness) {
      ER_HIP_TRY(hipMemsetAsync(g->d_fit, 0, (size_t)m * 2 * sizeof(double), g->stream));
      const int mx = max_points(i0, m, src);
      if (mx > 0) {
        const int bx = std::max(1, std::min(nblocks_of(mx), std::max(64, 8192 / m)));
        hipLaunchKernelGGL(k_fitness, dim3(bx, m), dim3(kBlock), 0, g->stream, g->d_pairs, g->d_state, (float)max_dist, g->d_fit);
        ER_HIP_TRY(hipGetLastError());
      }
      ER_HIP_TRY(hipMemcpyAsync(g->h_fit, g->d_fit, (size_t)m * 2 * sizeof(double), hipMemcpyDeviceToHost, g->stream));
      ER_HIP_TRY(hipStreamSynchronize(g->stream));
    }
    for (int q = 0; q < m; q++) {
      const IcpDev& st = g->h_state[q];
      memcpy(out + (size_t)(i0 + q) * 16, st.fin, 16 * sizeof(float));
      if (iterations) iterations[i0 + q] = st.iter;
      if (converged) converged[i0 + q] = st.conv ? 1 : 0;
      if (fitness) fitness[i0 + q] = g->h_fit[2 * q + 1] > 0 ? g->h_fit[2 * q] / g->h_fit[2 * q + 1] : DBL_MAX;
    }
  }
  return 0;
}

int er_icp_align(er_cloud_t src, er_cloud_t tgt, const float guess[16], double max_dist, int max_iter,
                 double transformation_epsilon, int stop_rule, float out[16], int* iterations, int* converged, double* fitness) {
  if (!guess || !out) return er::fail("er_icp_align: NULL argument");
  return er_icp_align_batch(1, &src, &tgt, guess, max_dist, max_iter, transformation_epsilon, stop_rule, out, iterations, converged, fitness);
}

int er_ransac_fitness_batch(er_cloud_t src, er_cloud_t tgt, int n_hyp, const float* M, float corr_dist_threshold, int* inliers,
                            double* fitness) {
  if (n_hyp < 0 || (n_hyp > 0 && (!M || !inliers))) return er::fail("er_ransac_fitness_batch: bad arguments");
  if (check_pair(src, tgt, (double)corr_dist_threshold, "er_ransac_fitness_batch")) return 1;
  if (n_hyp == 0) return 0;
  GroupLease L;
  if (L.acquire(src->device)) return 1;
  Group* w = L.g;
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

// Compaction chain of slots [s0, s0 + m) of the group (match already written): per-block counts (+ information terms), per-pair
// scan, stable compaction; the totals and the information terms come back to the pinned mirrors.  Records `done` on the stream.
static int corr_chain(Group* g, int s0, int m, int mxb, bool want_source, bool want_target, hipEvent_t done, hipStream_t S = nullptr,
                      hipStream_t compact_on = nullptr, hipEvent_t scanned = nullptr) {
  if (!S) S = g->stream;
  hipLaunchKernelGGL(k_count_blocks, dim3((mxb + kCountSlices - 1) / kCountSlices, m), dim3(kBlock), 0, S, g->d_pairs + s0, want_source ? 1 : 0,
                     want_target ? 1 : 0);
  hipLaunchKernelGGL(k_scan_blocks, dim3(m), dim3(1024), 0, S, g->d_pairs + s0, g->d_totals + s0, g->d_info + (size_t)s0 * kAcc,
                     (want_source || want_target) ? 1 : 0);
  ER_HIP_TRY(hipGetLastError());
  ER_HIP_TRY(hipMemcpyAsync(g->h_totals + s0, g->d_totals + s0, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, S));
  ER_HIP_TRY(hipMemcpyAsync(g->h_info + (size_t)s0 * kAcc, g->d_info + (size_t)s0 * kAcc, (size_t)m * kAcc * sizeof(double), hipMemcpyDeviceToHost, S));
  hipStream_t C = S;
  if (compact_on && compact_on != S && scanned) {               // the compaction (PCIe-bound when it writes into host memory) leaves the compute stream
    ER_HIP_TRY(hipEventRecord(scanned, S));
    ER_HIP_TRY(hipStreamWaitEvent(compact_on, scanned, 0));
    C = compact_on;
  }
  hipLaunchKernelGGL(k_compact, dim3(mxb, m), dim3(kBlock), 0, C, g->d_pairs + s0);
  ER_HIP_TRY(hipGetLastError());
  ER_HIP_TRY(hipEventRecord(done, C));
  return 0;
}

static int stage_reserve(Group* g, size_t ints) {
  if (ints <= g->stage_cap) return 0;
  ER_HIP_TRY(hipStreamSynchronize(g->copy_stream));
  ER_HIP_TRY(hipStreamSynchronize(g->copy_stream2));
  if (g->stage) (void)hipHostFree(g->stage);
  g->stage = nullptr;
  g->stage_cap = 0;
  ER_HIP_TRY(hipHostMalloc((void**)&g->stage, ints * sizeof(int), hipHostMallocDefault));
  g->stage_cap = ints;
  return 0;
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
  GroupLease L;
  if (L.acquire(src->device)) return 1;
  Group* g = L.g;
  if (group_describe(g, 0, 1, &src, &tgt, nullptr, true)) return 1;
  const int n = src->n, nb = nblocks_of(n);
  Mat12f M;
  for (int q = 0; q < 12; q++) M.m[q] = M16[q];
  ER_HIP_TRY(hipMemsetAsync(g->d_info, 0, kAcc * sizeof(double), g->stream));
  hipLaunchKernelGGL(k_ransac_match, dim3(nb), dim3(kBlock), 0, g->stream, g->d_pairs, M, corr_dist_threshold, corr_dist_threshold * corr_dist_threshold,
                     g->d_info);
  if (corr_chain(g, 0, 1, nb, info_source36 != nullptr, info_target36 != nullptr, g->ev)) return 1;
  ER_HIP_TRY(hipEventSynchronize(g->ev));
  const int total = g->h_totals[0];
  *n_inliers = total;
  if (info_source36) expand_information(g->h_info, info_source36);
  if (info_target36) expand_information(g->h_info + 10, info_target36);
  if (fitness && total > 0) *fitness = g->h_info[20] / (double)total;                               // :697-703
  const int ncopy = std::min(total, capacity);
  if (ncopy > 0) {
    const int kind = buffer_kind(pairs_host, src->device);
    if (kind < 0) return er::fail("er_ransac_inliers: the list buffer lives on another GPU than the clouds");
    const bool direct = kind >= 1;                              // page-locked host or device memory
    if (!direct && stage_reserve(g, (size_t)ncopy * 2)) return 1;
    ER_HIP_TRY(hipMemcpyAsync(direct ? pairs_host : g->stage, g->h_pairs[0].pairs, (size_t)ncopy * 2 * sizeof(int),
                              kind == 2 ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, g->stream));
    ER_HIP_TRY(hipStreamSynchronize(g->stream));
    if (!direct) memcpy(pairs_host, g->stage, (size_t)ncopy * 2 * sizeof(int));
  }
  return total > capacity ? er::fail("er_ransac_inliers: %d pairs exceed the capacity %d", total, capacity) : 0;
}

// FindCorrespondence of a pair list (CorresApp.cpp:112-210).  A group is described once; its pairs run in SUB-GROUPS of kCorrSub:
// search + compaction chain of sub-group s+1 are enqueued before the host waits for the totals of sub-group s, and the list copies
// of s (exactly total pairs each; PCIe-bound: 8 bytes per correspondence) run underneath on TWO copy streams in turn (one stream
// leaves ~12 us between consecutive copies: 38 instead of 53 GB/s, profiles/r03l_icp_timeline.txt).
constexpr int kCorrSub = 8;
int er_find_correspondence_batch(int n, const er_cloud_t* src, const er_cloud_t* tgt, const double* T, double dist, double normal_cos,
                                 int* const* pairs_host, const int* capacity, int* n_pairs, double* info36) {
  int device;
  if (n > 0 && (!T || !n_pairs || !pairs_host || !capacity)) return er::fail("er_find_correspondence: NULL argument");
  if (batch_prologue(n, src, tgt, dist, "er_find_correspondence", &device)) return 1;
  if (n == 0) return 0;
  size_t stage_need = 0;
  std::vector<char> direct((size_t)n, 0);                      // destination is page-locked: the list is copied straight into it
  // A list whose buffer the GPU can address and that has room for the largest possible list (one pair per source point) can be written THERE by k_compact:
  // no copy at all (direct_lists()).  The copies of the other lists are dealt to up to four copy streams: with two, both could land on ONE SDMA engine --
  // 83 MB of lists in 4.4 instead of 1.6 ms (20 against 51 GB/s), whole processes long, flipping after unrelated calls created more streams
  // (profiles/r06h_realistic_probe.txt; BENCH_r05's kinfu-like figure sat in the slow mode).
  std::vector<int*> written((size_t)n, nullptr);
  for (int i = 0; i < n; i++) {
    if (capacity[i] > 0 && !pairs_host[i]) return er::fail("er_find_correspondence: NULL pair buffer for pair %d", i);
    if (capacity[i] > 0) {
      void* view = nullptr;
      const int kind = buffer_kind(pairs_host[i], device, &view);              // (asked once per pair, not again at copy time)
      if (kind < 0) return er::fail("er_find_correspondence: the list buffer of pair %d lives on another GPU than its clouds", i);
      direct[(size_t)i] = (char)kind;                                           // 1: page-locked host, 2: device memory -- both are copied to directly
      if (!direct[(size_t)i]) stage_need += (size_t)std::min(capacity[i], src[i]->n) * 2;
      if (kind > 0 && view && capacity[i] >= src[i]->n && ((uintptr_t)view & 7u) == 0 && direct_lists() >= (kind == 2 ? 1 : 2) && !(kind == 1 && tl_lists_by_copy)) written[(size_t)i] = static_cast<int*>(view);
    }
  }
  GroupLease L;
  if (L.acquire(device)) return 1;
  Group* g = L.g;
  if (stage_need && stage_reserve(g, stage_need)) return 1;
  std::vector<long> staged((size_t)n, -1);
  size_t stage_used = 0;
  const int G = group_cfg();
  int rc = 0;
  const int ncs = list_copy_streams();
  for (int q = 0; q < 2 && q + 2 < ncs; q++)
    if (!g->copy_more[q]) ER_HIP_TRY(hipStreamCreateWithFlags(&g->copy_more[q], hipStreamNonBlocking));
  hipStream_t cs[4] = {g->copy_stream, ncs > 1 ? g->copy_stream2 : g->copy_stream, ncs > 2 ? g->copy_more[0] : g->copy_stream,
                       ncs > 3 ? g->copy_more[1] : (ncs > 1 ? g->copy_stream2 : g->copy_stream)};
  for (int i0 = 0; i0 < n; i0 += G) {
    const int m = std::min(G, n - i0);
    for (int q = 0; q < 4; q++) ER_HIP_TRY(hipStreamSynchronize(cs[q]));          // the previous group's lists have left the slabs
    if (group_describe(g, i0, m, src, tgt, T, true, written.data())) return 1;
    ER_HIP_TRY(hipMemsetAsync(g->d_info, 0, (size_t)m * kAcc * sizeof(double), g->stream));
    // sub-groups exist to put the list copies of one behind the kernels of the next (PCIe); when every list of this group stays in HBM there is nothing
    // to hide and the whole group is ONE sub-group: 4 launches instead of 4 per 8 pairs, no partly filled last waves in between
    bool all_dev = true;
    for (int q = 0; q < m; q++) all_dev = all_dev && (capacity[i0 + q] <= 0 || direct[(size_t)(i0 + q)] == 2 || written[(size_t)(i0 + q)]);
    // Sub-groups put list COPIES behind the next kernels.  Lists that k_compact writes into HOST memory cross PCIe with its stores (1.5 ms for the 83 MB of
    // a 50-pair list: the link's rate, scripts/ubench/d2h_paths.hip) while the searches need 0.7 ms of the GPU: the group is cut into ER_ICP_FC_SPLIT parts;
    // searches, counts and scans stay on the compute stream, the compactions go to two side streams behind an event each, so that one part's stores
    // leave while the next part searches.
    bool host_written = false;
    for (int q = 0; q < m; q++) host_written = host_written || (written[(size_t)(i0 + q)] && direct[(size_t)(i0 + q)] == 1);
    static const int fc_split = [] { const char* e = getenv("ER_ICP_FC_SPLIT"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : (v > 16 ? 16 : v); }();
    const bool split = all_dev && host_written && fc_split > 1 && m >= 2 * fc_split;
    const int sub = all_dev ? (split ? (m + fc_split - 1) / fc_split : std::max(m, 1)) : kCorrSub;
    const int nsub = (m + sub - 1) / sub;
    hipStream_t ks[2] = {g->stream, g->stream};
    while (split && (int)g->sub_ev2.size() < nsub) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return er::fail("er_find_correspondence: hipEventCreate failed");
      g->sub_ev2.push_back(e);
    }
    while ((int)g->sub_ev.size() < nsub) {
      hipEvent_t e = nullptr;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return er::fail("er_find_correspondence: hipEventCreate failed");
      g->sub_ev.push_back(e);
    }
    std::vector<hipEvent_t>& evs = g->sub_ev;
    auto cleanup = [&]() {};
    auto enqueue = [&](int s) -> int {
      const int s0 = s * sub, ms = std::min(sub, m - s0);
      int mxb = 1;
      for (int q = 0; q < ms; q++) mxb = std::max(mxb, g->h_pairs[s0 + q].nb);
      hipLaunchKernelGGL(k_find_corr, dim3(mxb, ms), dim3(kBlock), 0, ks[s & 1], g->d_pairs + s0, (float)dist, dist * dist, normal_cos);
      return corr_chain(g, s0, ms, mxb, info36 != nullptr, false, evs[(size_t)s], ks[s & 1], split ? ((s & 1) ? g->copy_stream2 : g->copy_stream) : nullptr,
                        split ? g->sub_ev2[(size_t)s] : nullptr);
    };
    if (split) {
      for (int s = 0; s < nsub; s++)
        if (enqueue(s)) { cleanup(); return 1; }
    } else if (enqueue(0)) { cleanup(); return 1; }
    for (int s = 0; s < nsub; s++) {
      if (!split && s + 1 < nsub && enqueue(s + 1)) { cleanup(); return 1; }
      const int s0 = s * sub, ms = std::min(sub, m - s0);
      if (hipEventSynchronize(evs[(size_t)s]) != hipSuccess) { cleanup(); return er::fail("er_find_correspondence: %s", hipGetErrorString(hipGetLastError())); }
      for (int q = 0; q < ms; q++) {
        const int i = i0 + s0 + q;
        const bool empty = src[i]->n == 0 || tgt[i]->n == 0;
        n_pairs[i] = empty ? 0 : g->h_totals[s0 + q];
        if (empty) g->h_totals[s0 + q] = 0;
        if (info36) {
          if (empty) memset(info36 + (size_t)i * 36, 0, 36 * sizeof(double));
          else expand_information(g->h_info + (size_t)(s0 + q) * kAcc, info36 + (size_t)i * 36);
        }
        if (n_pairs[i] > capacity[i]) rc = er::fail("er_find_correspondence: %d pairs exceed the capacity %d", n_pairs[i], capacity[i]);
      }
      // copies of this sub-group's lists on the copy stream (its kernels are done: evs[s] has been waited for)
      for (int q = 0; q < ncs; q++) ER_HIP_TRY(hipStreamWaitEvent(cs[q], evs[(size_t)s], 0));
      for (int q = 0; q < ms; q++) {
        const int i = i0 + s0 + q, ncopy = std::min(n_pairs[i], capacity[i]);
        if (ncopy <= 0 || written[(size_t)i]) continue;          // (written in place by k_compact)
        int* dst = pairs_host[i];
        if (!direct[(size_t)i]) {
          staged[(size_t)i] = (long)stage_used;
          dst = g->stage + stage_used;
          stage_used += (size_t)ncopy * 2;
        }
        if (hipMemcpyAsync(dst, g->h_pairs[s0 + q].pairs, (size_t)ncopy * 2 * sizeof(int), direct[(size_t)i] == 2 ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                           cs[i % ncs]) != hipSuccess) {
          cleanup();
          return er::fail("er_find_correspondence: %s", hipGetErrorString(hipGetLastError()));
        }
      }
    }
    cleanup();
  }
  for (int q = 0; q < 4; q++) ER_HIP_TRY(hipStreamSynchronize(cs[q]));
  for (int i = 0; i < n; i++)
    if (staged[(size_t)i] >= 0) memcpy(pairs_host[i], g->stage + staged[(size_t)i], (size_t)std::min(n_pairs[i], capacity[i]) * 2 * sizeof(int));
  return rc;
}

int er_find_correspondence(er_cloud_t src, er_cloud_t tgt, const double T[16], double dist, double normal_cos,
                           int* pairs_host, int capacity, int* n_pairs, double* info36) {
  if (!T || !n_pairs || (capacity > 0 && !pairs_host)) return er::fail("er_find_correspondence: NULL argument");
  int* const bufs[1] = {pairs_host};
  return er_find_correspondence_batch(1, &src, &tgt, T, dist, normal_cos, bufs, &capacity, n_pairs, info36);
}

// CCorresApp::Registration followed by CCorresApp::FindCorrespondence for a whole pair list in ONE call (CorresApp.cpp:212-319, then
// :112-210): per pair the pre-check count, the accept rule of :270 ("cnt >= reg_num_ || ( r1 > reg_ratio_ && r2 > reg_ratio_ )"), for
// the accepted pairs icp.align from the float32 cast of the guess (:295-312) and -- from the float64 cast of its result, as :312 stores
// it -- the correspondence list with the `Reduced too much` rule of :164-173 left to the caller (n_pairs and counts are both returned).
// The pairs are independent, so the list is cut into a few contiguous shares and every share runs the three stages on a host thread
// and a group workspace of its own: the host round trips of one share (accept rule, the states of an ICP chunk, list sizes) and its
// list copies over PCIe overlap the kernels of the others -- what the reference's "#pragma omp parallel for" does for its CPU loops.
// Results are those of the three *_batch calls in sequence (same kernels, same per-pair arithmetic; a share is a shorter list, so the slice count per
// workgroup of k_icp_iter -- and with it the grouping of the float64 sums -- can differ: last bits of a transform, see include/er_hip.h).
int er_registration_batch(int n, const er_cloud_t* src, const er_cloud_t* tgt, const double* T_guess, double reg_dist, int reg_num, double reg_ratio,
                          int max_iter, double transformation_epsilon, int stop_rule, double corr_dist, double normal_cos, int* counts,
                          int* accepted, float* T_final, int* iterations, int* converged, int* const* pairs_host, const int* capacity,
                          int* n_pairs, double* info36) {
  int device;
  if (n > 0 && (!T_guess || !counts || !accepted || !T_final || !pairs_host || !capacity || !n_pairs)) return er::fail("er_registration_batch: NULL argument");
  if (batch_prologue(n, src, tgt, reg_dist, "er_registration_batch", &device)) return 1;
  if (batch_prologue(n, src, tgt, corr_dist, "er_registration_batch", &device)) return 1;
  if (n == 0) return 0;
  const char* e = getenv("ER_ICP_SHARES");
  int shares = e ? atoi(e) : 6;                                              // (measured on the 50-pair list: 2 shares 10.5 k, 4: 10.4 k, 6: 10.9 k pairs/s)
  shares = std::max(1, std::min(shares, std::min(8, (n + 7) / 8)));          // at least ~8 pairs per share: every launch still fills the chip
  std::vector<int> rc((size_t)shares, 0);
  std::vector<std::string> why((size_t)shares);
  auto work = [&](int w) {
    const int lo = (int)((long)n * w / shares), hi = (int)((long)n * (w + 1) / shares), m = hi - lo;
    if (m <= 0) return;
    auto failed = [&]() { rc[(size_t)w] = 1; why[(size_t)w] = er_last_error(); };   // (the message lives in this thread's buffer)
    if (hipSetDevice(device) != hipSuccess) { er::fail("er_registration_batch: hipSetDevice(%d) failed", device); failed(); return; }
    if (er_icp_count_inliers_batch(m, src + lo, tgt + lo, T_guess + (size_t)lo * 16, reg_dist, counts + lo)) { failed(); return; }
    std::vector<int> idx;
    for (int i = lo; i < hi; i++) {
      const double r1 = (double)counts[i] / (double)tgt[i]->n, r2 = (double)counts[i] / (double)src[i]->n;   // :268-269 (0 / 0 = NaN: rejected)
      accepted[i] = (counts[i] >= reg_num || (r1 > reg_ratio && r2 > reg_ratio)) ? 1 : 0;
      n_pairs[i] = 0;
      if (iterations) iterations[i] = 0;
      if (converged) converged[i] = 0;
      for (int q = 0; q < 16; q++) T_final[(size_t)i * 16 + q] = (float)T_guess[(size_t)i * 16 + q];      // a rejected pair keeps its transform (:277-283)
      if (info36) memset(info36 + (size_t)i * 36, 0, 36 * sizeof(double));
      if (accepted[i]) idx.push_back(i);
    }
    const int a = (int)idx.size();
    if (a == 0) return;
    std::vector<er_cloud_t> s2((size_t)a), t2((size_t)a);
    std::vector<float> g2((size_t)a * 16), f2((size_t)a * 16);
    std::vector<double> T2((size_t)a * 16), I2(info36 ? (size_t)a * 36 : 0);
    std::vector<int> it2((size_t)a), cv2((size_t)a), cap2((size_t)a), np2((size_t)a);
    std::vector<int*> buf2((size_t)a);
    for (int k = 0; k < a; k++) {
      const int i = idx[(size_t)k];
      s2[(size_t)k] = src[i];
      t2[(size_t)k] = tgt[i];
      memcpy(&g2[(size_t)k * 16], &T_final[(size_t)i * 16], 16 * sizeof(float));                            // transformation_.cast<float>()
      cap2[(size_t)k] = capacity[i];
      buf2[(size_t)k] = pairs_host[i];
    }
    if (er_icp_align_batch(a, s2.data(), t2.data(), g2.data(), reg_dist, max_iter, transformation_epsilon, stop_rule, f2.data(), it2.data(), cv2.data(), nullptr)) {
      failed();
      return;
    }
    for (size_t q = 0; q < (size_t)a * 16; q++) T2[q] = (double)f2[q];                                        // getFinalTransformation().cast<double>()
    tl_lists_by_copy = shares > 1;
    const int frc = er_find_correspondence_batch(a, s2.data(), t2.data(), T2.data(), corr_dist, normal_cos, buf2.data(), cap2.data(), np2.data(),
                                                 info36 ? I2.data() : nullptr);
    tl_lists_by_copy = false;
    for (int k = 0; k < a; k++) {
      const int i = idx[(size_t)k];
      memcpy(&T_final[(size_t)i * 16], &f2[(size_t)k * 16], 16 * sizeof(float));
      if (iterations) iterations[i] = it2[(size_t)k];
      if (converged) converged[i] = cv2[(size_t)k];
      n_pairs[i] = np2[(size_t)k];
      if (info36) memcpy(info36 + (size_t)i * 36, &I2[(size_t)k * 36], 36 * sizeof(double));
    }
    if (frc) failed();
  };
  if (shares == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int w = 0; w < shares; w++) th.emplace_back(work, w);
    for (std::thread& t : th) t.join();
  }
  for (int w = 0; w < shares; w++)
    if (rc[(size_t)w]) return er::fail("%s", why[(size_t)w].c_str());
  return 0;
}

}  // extern "C"
