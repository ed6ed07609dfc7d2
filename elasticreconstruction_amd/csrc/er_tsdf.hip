// er_tsdf.hip -- path A of liber_hip.so: TSDF depth integration with control-grid warp on MI355X
// (gfx950).  Replaces the reference's TSDFVolume / TSDFVolumeUnit / ControlGrid / CIntegrateApp::Reproject
// (Integrate/TSDFVolume.cpp:19-132, TSDFVolumeUnit.cpp:4-21, ControlGrid.h:41-87, IntegrateApp.cpp:228-269).
//
// Data layout in HBM (one er_tsdf_s per GPU):
//   pool      float2[max_units][64][64][64]   {sdf_, weight_} interleaved per voxel, k fastest
//                                             (the reference keeps two float[64^3] per unit,
//                                             TSDFVolumeUnit.h:108-109; interleaving makes the
//                                             read-modify-write one 8-byte access per voxel)
//   ht_*      open-addressing hash map  hash_key -> {pool slot, 64-bit frame mask}; the device-side
//                                             twin of TSDFVolume::data_ (TSDFVolume.h:27)
//   lambda    float[rows*cols]                ScaleDepth's per-pixel ray-length factor (camera constant)
//   scaled    float[64][rows*cols]            scaled depth of every frame of the batch in flight
//   zbuf      uint32[64][rows*cols]           Reproject's z-buffer (atomicMin; 0xFFFFFFFF = empty); lastzero / zfix: its replay state
//
// Launch sequence for a batch of <= 64 frames (er_tsdf_integrate_frames); three batches are in flight (run_batch):
//  pre-pass stream (batch b on stream b mod 2):
//   k_reset        clears the frame masks of the slot's previous batch (deferred from the main stream)
//   [k_reproject_scatter -> k_reproject_fix]          per SOURCE pixel: warp + scatter-min   (A6/A7)
//   k_prepare      per pixel: ScaleDepth + unit key; marks bit f in the unit's frame mask,   (A3/A5)
//                  appends the unit to the batch list
//   k_plan         hands new units their pool slots, writes one record {key, slot, frame mask} per unit of the batch in cost order
//                  (frames in the mask) and resets the work queue k_integrate claims its items from
//  main stream:
//   k_integrate    per wave an 8 x 4 x 8 box of a unit: each voxel is loaded ONCE, run against every (A4)
//                  frame whose bit is set IN FRAME ORDER, stored once -> bit-identical to the reference's
//                  frame-by-frame loop with 1/batch of its HBM traffic
// All kernels are HBM/latency/VALU work on scattered voxels and pixels: no MFMA.
#include "er_common.h"
#include "er_tsdf_math.h"
#include "er_mc_table.h"

#include "../../include/er_hip.h"

#include <algorithm>
#include <cstdlib>
#include <utility>
#include <vector>

// Three translation units from this ONE source (round 4): the compiler flags that are best for the voxel pass are not the ones that
// are best for the pre-pass kernels (profiles/r04k_ab_compiler_flags.txt: without the SLP vectoriser's packed-math pairs -- which cost
// k_reproject_scatter 28 register moves per pixel -- and with the max-memory-clause scheduler the job gains 4.5 %; k_integrate alone is
// fastest with the max-ILP scheduler), and hipcc takes such flags per file.  er_tsdf_pre.hip and er_tsdf_int.hip include this file with
// ER_TSDF_TU = 1 (k_reproject_scatter, k_prepare) and 2 (k_integrate); the default, 0, is everything else: the other kernels and the
// host side.  The kernels that cross the boundary, and the structs in their signatures, live in a named namespace (external linkage);
// device helpers stay in the anonymous namespace and are compiled where they are used.
#ifndef ER_TSDF_TU
#define ER_TSDF_TU 0
#endif

namespace {

using namespace er;

constexpr int kBlock = 256;
constexpr int kEmptyKey = -1;
constexpr uint32_t kZEmpty = 0xFFFFFFFFu;

// counters[] slots
enum { C_NUNITS = 0, C_NBATCH = 1 /* and 6, 7: one per pipeline slot */, C_POOL_OVERFLOW = 2, C_TABLE_FULL = 3, C_OUT_OF_RANGE = 4,
       C_NBATCH1 = 6, C_NBATCH2 = 7, C_ZERO_WRITE = 8 /* 8, 9: frames 0-31 / 32-63 of the batch; 10, 11 for the second pre-pass stream */,
       C_ZERO_WRITE1 = 10, C_COUNT = 12 };
constexpr int kDepth = 3;                // batches in flight: voxel pass of n, pre-passes of n+1 and n+2 (depth 2 with one pre-pass
constexpr int kAux = 2;                  // stream = the round-1 pipeline: profiles/r02n_ab_pipeline_depth_hw_queues.txt); pre-pass streams:
                                         // batch b runs on stream b mod kAux
constexpr int kNbatchSlot[3] = {C_NBATCH, C_NBATCH1, C_NBATCH2};
constexpr int kZeroFlagSlot[2] = {C_ZERO_WRITE, C_ZERO_WRITE1};

__device__ __forceinline__ unsigned hash_unit_key(int key, int shift) { return ((unsigned)key * 2654435761u) >> shift; }

// Lock-free find-or-insert.  The entry index is stable, so callers never wait for anybody.
__device__ int ht_find_or_insert(int* __restrict__ ht_key, int cap_mask, int shift, int key) {
  unsigned h = hash_unit_key(key, shift);
  for (int probe = 0; probe <= cap_mask; ++probe) {
    int e = (int)((h + (unsigned)probe) & (unsigned)cap_mask);
    int k = __hip_atomic_load(&ht_key[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == key) return e;
    if (k == kEmptyKey) {
      int old = atomicCAS(&ht_key[e], kEmptyKey, key);
      if (old == kEmptyKey || old == key) return e;
    }
  }
  return -1;
}

#if ER_TSDF_TU == 0
// ------------------------------------------------------------------------------------------------
// ScaleDepth's camera-constant factor (TSDFVolume.cpp:24-26), tabulated once per volume.
__global__ void k_lambda(float* __restrict__ lambda, int cols, int rows, Camera cam) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= cols * rows) return;
  lambda[p] = scale_lambda(p % cols, p / cols, cam);
}

// Stand-alone ScaleDepth (TSDFVolume.cpp:19-36) for er_tsdf_scale_depth.
__global__ void k_scale_depth(const uint16_t* __restrict__ depth, const float* __restrict__ lambda,
                              float* __restrict__ scaled, int pixels, float itrunc) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= pixels) return;
  scaled[p] = scale_depth_px(depth[p], lambda[p], itrunc);
}

#endif  // ER_TSDF_TU == 0
// ------------------------------------------------------------------------------------------------
// Reproject, IntegrateApp.cpp:247-268: every source pixel is warped through its fragment's control
// grid and scattered into the frame's z-buffer.  The reference's sequential "write if empty or
// closer" is an order-independent min for dd != 0; a write of dd == 0 RESETS the cell (0 means
// empty), which is order dependent.  Such a write (a warped depth below 0.5 mm: practically never) records its source index
// in lastzero (max) and raises the FRAME's bit in the stream's flag word; k_reproject_fix then scatters the flagged frames a
// second time into a side buffer, zfix, under the replay rule -- only writes that come after the cell's last zero write
// count -- and the consumer of the z-buffer (k_prepare / k_zbuf_to_depth) takes cells with lastzero > 0 from zfix.
}  // namespace
namespace er_tsdf_k {
using namespace er;
struct ReprojArgs {
  const uint16_t* depth;
  int n_frames, cols, rows;
  Camera cam;
  CameraInv cami;
  const double* seg12;
  const double* madj12;
  const int* grid_index;
  const float* ctr;
  int res;
  float grid_ul;
  int floats_per_grid;
  uint32_t* zbuf;
  uint32_t* lastzero;
  uint32_t* zfix;                        // the replay's z-buffer (all-empty outside a replay; re-armed by the consumer)
  int* zero_flag;                        // int[2], bit f: frame f of the batch saw a write of dd == 0 (one pair per pre-pass stream)
};
__global__ void k_reproject_scatter(ReprojArgs A);
}  // namespace er_tsdf_k
namespace {
using namespace er_tsdf_k;

// The write half of one source pixel p of frame f that landed on `cell` with depth dd (IntegrateApp.cpp:260-263).
__device__ __forceinline__ void scatter_px(const ReprojArgs& A, int f, int p, int cell, uint16_t dd, int replay) {
  const size_t o = (size_t)f * ((size_t)A.cols * A.rows) + cell;
  if (!replay) {
    if (dd != 0) {
      atomicMin(&A.zbuf[o], (uint32_t)dd);
    } else {
      atomicMax(&A.lastzero[o], (uint32_t)p + 1u);
      atomicOr(&A.zero_flag[f >> 5], 1 << (f & 31));
    }
  } else {
    const uint32_t lz = A.lastzero[o];
    if (dd != 0 && lz > 0 && (uint32_t)p + 1u > lz) atomicMin(&A.zfix[o], (uint32_t)dd);
  }
}

// One source pixel (u, v) of frame f through the EXACT chain: warp, then scatter (replay = 0) or re-scatter under the
// replay rule (replay = 1).
__device__ __forceinline__ void reproject_scatter_px(const ReprojArgs& A, int f, int u, int v, int replay) {
  const int pixels = A.cols * A.rows;
  const int p = v * A.cols + u;
  const uint16_t d = A.depth[(size_t)f * pixels + p];
  if (d == 0) return;                                                   // UVD2XYZ false
  int cell;
  uint16_t dd;
  if (!reproject_px(u, v, d, A.cam, A.cami, A.cols, A.rows, A.seg12 + f * 16, A.madj12 + f * 12,
                    A.ctr + (size_t)A.grid_index[f] * A.floats_per_grid, A.res, A.grid_ul, cell, dd))
    return;
  scatter_px(A, f, p, cell, dd, replay);
}

}  // namespace
#if ER_TSDF_TU == 1
namespace er_tsdf_k {
__global__ void k_reproject_scatter(ReprojArgs A) {
  // 64 x 4 pixel tiles per 256-thread workgroup, frame = blockIdx.z: no integer divisions for the indices.
  const int f = blockIdx.z;
  const int u = blockIdx.x * 64 + (threadIdx.x & 63);
  const int v = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (u >= A.cols || v >= A.rows) return;
  reproject_scatter_px(A, f, u, v, 0);
}
}  // namespace er_tsdf_k
#endif  // ER_TSDF_TU == 1
namespace {

#if ER_TSDF_TU == 0
// The order-dependent case (a write of dd == 0 resets the cell: "0 means empty"), replayed exactly.  ONE launch of kFixBlocks
// single-wave workgroups that return at once unless a frame of the batch is flagged -- practically never -- and otherwise share the
// pixels of the flagged frames: every source pixel is warped again and scattered into zfix under the replay rule.  One phase, no
// ordering between workgroups; the consumer merges (take_z below) and re-arms lastzero / zfix, k_plan (or the single-frame entry
// point) clears the flag words.  Single-wave workgroups with a capped register budget: they find a free wave slot at once next to the
// persistent k_integrate workgroups (a 256-thread workgroup waited 46 us on average for four slots on one CU).
constexpr int kFixThreads = 64;
constexpr int kFixBlocks = 256;
__global__ __launch_bounds__(kFixThreads) __attribute__((amdgpu_num_vgpr(48))) void k_reproject_fix(ReprojArgs A) {
  const int fl0 = A.zero_flag[0], fl1 = A.zero_flag[1];
  if ((fl0 | fl1) == 0) return;
  const int pixels = A.cols * A.rows;
  for (int f = 0; f < A.n_frames; f++) {
    if ((((f < 32 ? fl0 : fl1) >> (f & 31)) & 1) == 0) continue;
    for (int p = blockIdx.x * kFixThreads + threadIdx.x; p < pixels; p += gridDim.x * kFixThreads) reproject_scatter_px(A, f, p % A.cols, p / A.cols, 1);
  }
}

#endif  // ER_TSDF_TU == 0
// The consumer's half of the replay: the value of z-buffer cell o of a FLAGGED frame (z = what the plain scatter-min left there);
// cells that saw a zero write take the replay's value and re-arm both side buffers.
__device__ __forceinline__ uint32_t take_z(uint32_t z, size_t o, uint32_t* __restrict__ lastzero, uint32_t* __restrict__ zfix) {
  if (lastzero[o] == 0) return z;
  const uint32_t r = zfix[o];
  zfix[o] = kZEmpty;
  lastzero[o] = 0;
  return r;
}

#if ER_TSDF_TU == 0
__global__ void k_zbuf_to_depth(uint32_t* __restrict__ zbuf, uint16_t* __restrict__ depth, long total, uint32_t* __restrict__ lastzero,
                                uint32_t* __restrict__ zfix, const int* __restrict__ zero_flag) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  uint32_t z = zbuf[t];
  zbuf[t] = kZEmpty;                                                    // leave the z-buffer re-armed
  if (zero_flag[0] & 1) z = take_z(z, (size_t)t, lastzero, zfix);       // (single frame: bit 0)
  depth[t] = (z == kZEmpty) ? (uint16_t)0 : (uint16_t)z;
}
#endif  // ER_TSDF_TU == 0

// ------------------------------------------------------------------------------------------------
// Per pixel of every frame of the batch: ScaleDepth (TSDFVolume.cpp:19-36) and the unit-touch half
// of TSDFVolume::Integrate (TSDFVolume.cpp:45-58).
//
// One 1024-thread workgroup owns a 32x32 pixel TILE of one frame (a 64^3 unit projects to >100x100
// pixels at room scale, so a tile nearly always sees 1-3 units).  Lanes whose key differs from their
// left neighbour's append it to a small LDS list; after the barrier the list is de-duplicated and only
// the DISTINCT keys of the tile go to the global hash map.  Without this every wave hammered the same
// few hash entries with device-scope atomics at the same moment (measured: 90 % of wave time waiting).
// Frames of the scaled-depth buffer are kScaledPad floats apart beyond their pixels; the pad stays 0.0f for ever (zero-filled at
// create, never written): a voxel whose projection misses the image gathers from it instead of taking a predicated load.
#ifndef ER_FRAMES_TRANSPOSED
#define ER_FRAMES_TRANSPOSED 1           // k_integrate's culling reads the frame constants component-major (Staging::fxT); 0: frames[lane]
#endif
#ifndef ER_TILE_FRAME_FASTEST
#define ER_TILE_FRAME_FASTEST 1          // tile_max / tile_lo / tile_lo_fine as [tile][frame of the batch] (0: [frame][tile], rounds 1-5)
#endif
constexpr int kScaledPad = 64;
constexpr int kTile = 32;
// Granularity of tile_lo_fine, the second-level per-tile MINIMUM of the scaled depth behind k_integrate's "full" verdict: 2^kLoShift pixels.
// 16-pixel tiles next to the 32-pixel tiles of tile_max / tile_lo: a pixel without usable depth (the warp's scatter leaves holes) spoils the
// minimum of its whole tile.  The fine tiles are the SECOND level of the verdict (er_tsdf_math.h: patch_may_update_box): the 32-pixel minimum
// decides first, the fine ones are read only when it fails for a patch that lies clearly in front of everything under it.
#ifndef ER_TILE_LO_SHIFT
#define ER_TILE_LO_SHIFT 4
#endif
constexpr int kLoShift = ER_TILE_LO_SHIFT;
static_assert(kLoShift >= 3 && kLoShift <= 5, "tile_lo tiles of 8, 16 or 32 pixels");
constexpr int kLoSub = kTile >> kLoShift;               // tile_lo tiles per side of a 32 x 32 k_prepare tile: 4, 2 or 1
constexpr int kTileKeys = 96;

// Marks frame f in the unit's mask; the first toucher of the unit IN THIS BATCH (unique: its atomicOr
// returned 0) appends the unit to the batch list.  The pool slot of a unit that is new to the volume is handed out
// by k_plan (unit_slot_acquire below), on the same pre-pass stream.
//
// Unit-shard mode (the bit-exact multi-GPU alternative): with shard.y > 1 GPUs every GPU runs the pre-pass of
// ALL frames but only owns -- allocates, integrates, reports -- the units with unit_owner(key) == shard.x.  Units are
// disjoint (TSDFVolume.cpp:45-63) and each one still sees every frame in order, so the union over the GPUs equals the
// single-GPU volume bit for bit; no collective touches the volume.
__device__ void touch_unit(int key, int f, int* __restrict__ ht_key, int* __restrict__ ht_slot,
                           unsigned long long* __restrict__ ht_mask, int cap_mask, int hash_shift,
                           int* __restrict__ batch, int* __restrict__ nbatch, int* __restrict__ counters, int2 shard) {
  if (shard.y > 1 && unit_owner(key, shard.y) != shard.x) return;
  const int e = ht_find_or_insert(ht_key, cap_mask, hash_shift, key);
  if (e < 0) {
    atomicOr(&counters[C_TABLE_FULL], 1);
    return;
  }
  const unsigned long long bit = 1ull << f;
  const unsigned long long seen = __hip_atomic_load(&ht_mask[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (seen & bit) return;                                               // touched_unit.find, TSDFVolume.cpp:53
  const unsigned long long old = atomicOr(&ht_mask[e], bit);
  if (old != 0ull) return;
  batch[atomicAdd(nbatch, 1)] = e;
}

constexpr int kPrepThreads = 256;                       // 32 x 8 threads, 4 pixel rows each
constexpr int kPrepPix = kTile * kTile / kPrepThreads;  // pixels per thread

}  // namespace
namespace er_tsdf_k {
__global__ __launch_bounds__(kPrepThreads) void k_prepare(
    const uint16_t* __restrict__ depth, uint32_t* __restrict__ zbuf, int n_frames, int cols, int rows,
    Camera cam, CameraInv cami, const float* __restrict__ lambda, const double* __restrict__ T12, float* __restrict__ scaled,
    int* __restrict__ ht_key, int* __restrict__ ht_slot, unsigned long long* __restrict__ ht_mask, int cap_mask,
    int hash_shift, int* __restrict__ batch, int* __restrict__ nbatch,
    int* __restrict__ counters, float* __restrict__ tile_max, float* __restrict__ tile_lo, float* __restrict__ tile_lo_fine, int2 shard,
    uint32_t* __restrict__ lastzero, uint32_t* __restrict__ zfix, const int* __restrict__ zero_flag);
}  // namespace er_tsdf_k
#if ER_TSDF_TU == 1
namespace er_tsdf_k {
__global__ __launch_bounds__(kPrepThreads) void k_prepare(
    const uint16_t* __restrict__ depth, uint32_t* __restrict__ zbuf, int n_frames, int cols, int rows,
    Camera cam, CameraInv cami, const float* __restrict__ lambda, const double* __restrict__ T12, float* __restrict__ scaled,
    int* __restrict__ ht_key, int* __restrict__ ht_slot, unsigned long long* __restrict__ ht_mask, int cap_mask,
    int hash_shift, int* __restrict__ batch, int* __restrict__ nbatch,
    int* __restrict__ counters, float* __restrict__ tile_max, float* __restrict__ tile_lo, float* __restrict__ tile_lo_fine, int2 shard,
    uint32_t* __restrict__ lastzero, uint32_t* __restrict__ zfix, const int* __restrict__ zero_flag) {
  __shared__ int s_keys[kTileKeys];
  __shared__ int s_n;
  __shared__ float s_wmax[kPrepThreads / 64], s_wlo[kLoSub][kLoSub][kPrepThreads / 64];
  const int pixels = cols * rows;
  const int f = blockIdx.z;
  const bool replayed = zbuf && ((zero_flag[f >> 5] >> (f & 31)) & 1);  // this frame saw a zero write (uniform; practically never)
  const int tx = threadIdx.x & (kTile - 1), ty = threadIdx.x >> 5;      // ty in [0, 8)
  const int x = blockIdx.x * kTile + tx;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  // Each thread owns 4 pixels of its column (rows ty, ty+8, ty+16, ty+24 of the tile): the four loads are
  // issued together, which is what hides the HBM/L2 latency here (the kernel is latency-, not VALU-bound).
  uint16_t d[kPrepPix];
  float lam[kPrepPix];
#pragma unroll
  for (int q = 0; q < kPrepPix; q++) {
    const int y = blockIdx.y * kTile + ty + q * (kTile / kPrepPix);
    d[q] = 0;
    lam[q] = 0.0f;
    if (x < cols && y < rows) {
      const int p = y * cols + x;
      const size_t o = (size_t)f * pixels + p;
      if (zbuf) {
        uint32_t z = zbuf[o];
        zbuf[o] = kZEmpty;                                              // re-arm the z-buffer for the next batch
        if (replayed) z = take_z(z, o, lastzero, zfix);
        d[q] = (z == kZEmpty) ? (uint16_t)0 : (uint16_t)z;
      } else {
        d[q] = depth[o];
      }
      lam[q] = lambda[p];
    }
  }
  float wmax = 0.0f, vlo[kPrepPix];
#pragma unroll
  for (int q = 0; q < kPrepPix; q++) vlo[q] = 3.0e38f;
#pragma unroll
  for (int q = 0; q < kPrepPix; q++) {
    const int y = blockIdx.y * kTile + ty + q * (kTile / kPrepPix);
    int key = -1;
    if (x < cols && y < rows) {
      const float sc = scale_depth_px(d[q], lam[q], cam.integration_trunc);
      scaled[(size_t)f * (pixels + kScaledPad) + y * cols + x] = sc;
      wmax = fmaxf(wmax, sc);
      vlo[q] = sc > 0.001f ? sc : 0.0f;                                 // (min over EVERY pixel of its tile below: 0 as soon as one carries no usable depth
                                                                        // (a NaN depth -- degenerate camera -- fails ":82 dp > 0.001" too: it counts as 0, fminf alone would skip it)
      if (d[q] > 0) {                                                   // TSDFVolume.cpp:47 (no range cut-off)
        key = touch_key(x, y, d[q], cam, cami, T12 + f * 12);
        if (key < 0) atomicAdd(&counters[C_OUT_OF_RANGE], 1);
      }
    }
    const int left = __shfl_up(key, 1);
    const bool leader = key >= 0 && (tx == 0 || left != key);
    if (leader) {
      const int slot = atomicAdd(&s_n, 1);
      if (slot < kTileKeys) {
        s_keys[slot] = key;
      } else {                                                          // list full (pathological tile): go direct
        touch_unit(key, f, ht_key, ht_slot, ht_mask, cap_mask, hash_shift, batch, nbatch, counters, shard);
      }
    }
  }
  // max of the scaled depth over the 32 x 32 tile and min over its kLoSub x kLoSub sub-tiles of 2^kLoShift pixels (consumed by
  // patch_may_update_box in k_integrate: culling / the full verdict).  A thread's pixel q lies in row 8 q + ty of the tile, column tx: the
  // sub-tile row is (8 q + ty) >> kLoShift, the column tx >> kLoShift; a wave holds rows ty = 2 w, 2 w + 1 (lane = 32 (ty & 1) + tx).
  for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off));
  if ((threadIdx.x & 63) == 0) s_wmax[threadIdx.x >> 6] = wmax;
  {
    constexpr int qper = kPrepPix / kLoSub;             // pixel rows q of a thread per sub-tile row: 1, 2 or 4
#pragma unroll
    for (int sr = 0; sr < kLoSub; sr++) {
      float r = vlo[sr * qper];
#pragma unroll
      for (int e = 1; e < qper; e++) r = fminf(r, vlo[sr * qper + e]);
#pragma unroll
      for (int off = 1; off < (1 << kLoShift); off <<= 1) r = fminf(r, __shfl_xor(r, off));     // the columns of the sub-tile
      r = fminf(r, __shfl_xor(r, 32));                                                        // the wave's two rows
      if ((threadIdx.x & 32) == 0 && (tx & ((1 << kLoShift) - 1)) == 0) s_wlo[sr][tx >> kLoShift][threadIdx.x >> 6] = r;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = 0.0f, lo = 3.0e38f;
    for (int w = 0; w < kPrepThreads / 64; w++) m = fmaxf(m, s_wmax[w]);
    for (int e = 0; e < kLoSub * kLoSub * (kPrepThreads / 64); e++) lo = fminf(lo, (&s_wlo[0][0][0])[e]);
#if ER_TILE_FRAME_FASTEST
    const size_t t = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * ER_MAX_BATCH + f;      // [tile][frame]: see k_integrate's culling
#else
    const size_t t = ((size_t)f * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#endif
    tile_max[t] = m;
    tile_lo[t] = lo;                                      // the 32-pixel minimum: first level of the full verdict
  }
  if (kLoShift < 5 && (int)threadIdx.x < kLoSub * kLoSub) {
    const int sr = threadIdx.x / kLoSub, sg = threadIdx.x % kLoSub;
    float lo = 3.0e38f;
    for (int w = 0; w < kPrepThreads / 64; w++) lo = fminf(lo, s_wlo[sr][sg][w]);
    const int lx = blockIdx.x * kLoSub + sg, ly = blockIdx.y * kLoSub + sr;
    const int lo_tx = (cols + (1 << kLoShift) - 1) >> kLoShift, lo_ty = (rows + (1 << kLoShift) - 1) >> kLoShift;
#if ER_TILE_FRAME_FASTEST
    if (lx < lo_tx && ly < lo_ty) tile_lo_fine[((size_t)ly * lo_tx + lx) * ER_MAX_BATCH + f] = lo;
#else
    if (lx < lo_tx && ly < lo_ty) tile_lo_fine[((size_t)f * lo_ty + ly) * lo_tx + lx] = lo;
#endif
  }
  const int n = min(s_n, kTileKeys);
  if ((int)threadIdx.x < n) {
    const int k = s_keys[threadIdx.x];
    bool dup = false;
    for (int j = 0; j < (int)threadIdx.x; j++) dup = dup || (s_keys[j] == k);
    if (!dup) touch_unit(k, f, ht_key, ht_slot, ht_mask, cap_mask, hash_shift, batch, nbatch, counters, shard);
  }
}
}  // namespace er_tsdf_k
#endif  // ER_TSDF_TU == 1
namespace {

// ------------------------------------------------------------------------------------------------
// Work plan of one batch (single workgroup; a batch touches at most a few hundred units): the units of the
// batch list sorted by DESCENDING cost = popcount(frame mask) -- the order in which the persistent workgroups of k_integrate
// claim their items from the work queue (longest-processing-time first) -- and the queue head reset to 0.
constexpr int kRows = 4;                  // register rows per lane of k_integrate: a wave owns an 8 x 4 x 8 box of voxels (2 x 4 x 8 lanes x 4 rows)
constexpr int kItemsPerUnit = 256;       // work items per unit: 8 x 8 x 16 voxels per 256-thread workgroup

}  // namespace
namespace er_tsdf_k {
struct Plan {
  int n_units;
  int next;      // work queue of k_integrate: index of the next unclaimed item (reset by k_plan)
};

// What k_integrate needs to know about one unit of the batch, in ONE 16-byte scalar load (round 3; it used to chase plan entry ->
// hash key -> pool slot -> frame mask through four dependent loads per item).
struct PlanRec {
  int key;                  // hash_key of the unit (TSDFVolume.h:62-64)
  int slot;                 // pool slot; < 0: the pool is exhausted (reported by the host), the unit is skipped
  unsigned long long mask;  // frames of the batch that touch the unit
};
}  // namespace er_tsdf_k
namespace {

#if ER_TSDF_TU == 0
// Pool slot of hash entry e; hands the slot out on the unit's first ever visit (data_.find( key ) == end, TSDFVolume.cpp:55; pool
// memory is zero-filled up front).  Called by ONE thread per unit from k_plan.  The pre-passes of two batches run concurrently, so
// two k_plan launches can race for a new unit: one wins the compare-and-swap (-1 -> -2), draws the slot and publishes it; the
// other polls until it appears (the winner is a running thread of a resident single-workgroup kernel, so the wait is bounded).
// -3 = pool exhausted.  (Rounds 1-2 did this from k_integrate, per wave and item.)
// Two phases in program order -- winners draw and publish WITHOUT ever waiting, only then do the losers poll -- so that lanes of one
// wave that lost against the other launch cannot hold up lanes of the same wave that won (a wave runs the two sides of a divergent
// branch one after the other).
__device__ int unit_slot_acquire(int e, int key, int* __restrict__ ht_slot, int* __restrict__ unit_key, int max_units, int* __restrict__ counters) {
  int slot = __hip_atomic_load(&ht_slot[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool won = false;
  if (slot == -1) won = atomicCAS(&ht_slot[e], -1, -2) == -1;
  if (won) {
    const int s = atomicAdd(&counters[C_NUNITS], 1);
    if (s < max_units) {
      unit_key[s] = key;
      __threadfence();
      slot = s;
    } else {
      atomicOr(&counters[C_POOL_OVERFLOW], 1);
      slot = -3;
    }
    atomicExch(&ht_slot[e], slot);
  }
  if (!won && (slot == -1 || slot == -2)) {
    do {
      slot = __hip_atomic_load(&ht_slot[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (slot == -2) __builtin_amdgcn_s_sleep(8);
    } while (slot == -2);
  }
  return slot;
}

__global__ __launch_bounds__(256) void k_plan(const int* __restrict__ batch, const int* __restrict__ nbatch,
                                              const unsigned long long* __restrict__ ht_mask, const int* __restrict__ ht_key,
                                              int* __restrict__ ht_slot, int* __restrict__ unit_key, int max_units,
                                              int* __restrict__ counters, PlanRec* __restrict__ plan_rec, Plan* __restrict__ plan,
                                              int* __restrict__ zero_flag) {
  __shared__ int hist[65];
  __shared__ int start[66];
  const int n = *nbatch;                            // <= hash capacity = size of plan_rec
  for (int t = threadIdx.x; t < 65; t += blockDim.x) hist[t] = 0;
  __syncthreads();
  for (int t = threadIdx.x; t < n; t += blockDim.x) atomicAdd(&hist[__popcll(ht_mask[batch[t]])], 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    int acc = 0;
    for (int c = 64; c >= 0; c--) {                 // descending cost
      start[c] = acc;
      acc += hist[c];
    }
    plan->n_units = n;
    plan->next = 0;
    if (zero_flag) zero_flag[0] = zero_flag[1] = 0;     // Reproject's replay flags of this batch: consumed by the k_prepare in front of this launch
  }
  __syncthreads();
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const int e = batch[t];
    const unsigned long long mask = ht_mask[e];
    const int key = ht_key[e];
    PlanRec r;
    r.key = key;
    r.slot = unit_slot_acquire(e, key, ht_slot, unit_key, max_units, counters);
    r.mask = mask;
    plan_rec[atomicAdd(&start[__popcll(mask)], 1)] = r;                 // position in descending cost order
  }
}
#endif  // ER_TSDF_TU == 0

// ------------------------------------------------------------------------------------------------
// IntegrateVolumeUnit (TSDFVolume.cpp:69-102) for every touched unit of the batch.
// Work item = 1024 voxels of a unit for one 256-thread workgroup (256 items per unit); each wave owns 256 of them in kRows = 4 register rows of 64 -- a
// 8 x 4 x 8 box (mapping below).  The voxels stay in registers while the wave walks the unit's frame mask in ASCENDING frame order (wave-uniform loop:
// the frame constants arrive by scalar loads) -- per voxel exactly the reference's frame-by-frame sequence.
// Items come from ONE global work queue in cost order (k_plan), claimed when the workgroup is free.  (Static deals, per-XCD queues and look-ahead
// claims were all measured slower: profiles/HISTORY.md "Path A: the schedule of k_integrate".)
#ifndef ER_INT_LANE_SHAPE
#define ER_INT_LANE_SHAPE 1              // lanes of a wave of k_integrate: 1 = 2 x 4 x 8 voxels (ships), 0 = 1 x 8 x 8 (rounds 2-5)
#endif
#ifndef ER_INT_MIN_BLOCKS
#define ER_INT_MIN_BLOCKS 5
#endif
constexpr int kIntMinBlocks = ER_INT_MIN_BLOCKS;          // register budget handed to the compiler: 5 workgroups of 4 waves per CU = 102 VGPRs (the kernel uses 93 with 3, 4
                                          // or 5; with 6 it spills).  Same instructions, another register assignment: +1.0 % on the job against 4, five
                                          // interleaved runs out of five (profiles/r06v_ab_min_blocks.txt).  The grid launches ER_INT_BLOCKS_PER_CU = 3
                                          // workgroups per CU -- the free registers go to the co-running pre-pass kernels
// kSure: the square-root-free "sure" path of the frame loop (voxel_classify needs dp < 64 m; the host picks the instantiation
// from integration_trunc, which bounds every scaled depth).
}  // namespace
namespace er_tsdf_k {
template <bool kSure>
__global__ __launch_bounds__(kBlock, kIntMinBlocks) void k_integrate(
    float2* __restrict__ pool, const PlanRec* __restrict__ plan_rec, Plan* __restrict__ plan,
    const FrameXform* __restrict__ frames, const float* __restrict__ scaled, const float* __restrict__ tile_max,
    const float* __restrict__ tile_lo, const float* __restrict__ tile_lo_fine, int tiles_x, int tiles_y, Camera cam, int cols, int rows);
}  // namespace er_tsdf_k
#if ER_TSDF_TU == 2
namespace er_tsdf_k {
template <bool kSure>
__global__ __launch_bounds__(kBlock, kIntMinBlocks) void k_integrate(
    float2* __restrict__ pool, const PlanRec* __restrict__ plan_rec, Plan* __restrict__ plan,
    const FrameXform* __restrict__ frames, const float* __restrict__ scaled, const float* __restrict__ tile_max,
    const float* __restrict__ tile_lo, const float* __restrict__ tile_lo_fine, int tiles_x, int tiles_y, Camera cam, int cols, int rows) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int pixels = cols * rows;
  const int lo_tiles_x = (cols + (1 << kLoShift) - 1) >> kLoShift;
#if !ER_TILE_FRAME_FASTEST
  const int lo_tiles = lo_tiles_x * ((rows + (1 << kLoShift) - 1) >> kLoShift);
#endif
  const int n_items = plan->n_units * kItemsPerUnit;
  // Work queue: the items are sorted by descending cost (k_plan) and every workgroup claims the next one when it is done with its own (one atomic per
  // item and workgroup; 3 persistent workgroups per CU): longest-processing-time-first.  The culling and the full / sure shortcuts make the real cost of
  // an item unpredictable; with a static deal the kernel lasted as long as its unluckiest workgroup.  Two barriers per item on purpose: they keep the
  // four boxes of an item -- neighbours in the volume, hence in every depth image -- in step on one CU (barrier-free hand-outs measured 1.7 x the time
  // per frame visit, profiles/r05m_*).
  __shared__ int s_item;
  for (;;) {
    __syncthreads();                                                     // everybody is done with the previous s_item
    if (threadIdx.x == 0) s_item = atomicAdd(&plan->next, 1);
    __syncthreads();
    const int item = s_item;
    if (item >= n_items) break;
    const PlanRec rec = plan_rec[item >> 8];                              // (wave-uniform: one 16-byte scalar load)
    // The wave owns a COMPACT 8 x 4 x 8 BOX of the unit.  Lanes = 2 slabs x 4 x 8 voxels (il, jl, kl), register row r = the next pair of slabs
    // (i = i0 + il + 2 r); the workgroup's item = 8 x 8 x 16 voxels (waves: 2 along j, 2 along k).  Why this shape: a depth gather costs the vector L1
    // ~0.6 clocks per DISTINCT address (profiles/r06y_gather_rates.txt) and k_integrate lives on its gathers (every gather issued twice: -19 % frames/s,
    // profiles/r06z_ab_lane_shape.txt); the 64 voxels of an 8 x 8 plane -- rounds 2-5: lanes = one slab, rows = 4 slabs -- project onto 64 distinct pixels
    // seen face-on, a 2 x 4 x 8 block onto fewer from every direction.  +4 % on the job against the 1 x 8 x 8 lanes (ER_INT_LANE_SHAPE 0, kept for that
    // comparison); 4 x 4 x 4, 2 x 8 x 4, 2 x 2 x 16, 4 x 2 x 8, 1 x 4 x 16 lanes and two other item shapes measured behind it.  A compact box keeps a tight
    // pixel hull (culling, the "inside" verdict), few idle lanes at surfaces and frustum borders and few patches that cross a surface; four rows per lane
    // keep the longest items short and the kernel at 93 VGPRs.  Voxel accesses: eight 8-byte voxels = one 64-byte segment per (il, jl).
#if ER_INT_LANE_SHAPE == 0
    const int ilane = 0;
    const int i = ((item >> 4) & 15) * 4;
    const int j0 = ((item >> 2) & 3) * 16 + (wave >> 1) * 8;
    const int jlane = lane >> 3, k0 = (item & 3) * 16 + (wave & 1) * 8, klane = lane & 7;
    constexpr int jspan = 8, kspan = 8, ispan = kRows, istep = 1;
#else
    const int ilane = lane >> 5;
    const int i = ((item >> 5) & 7) * 8 + ilane;
    const int j0 = ((item >> 2) & 7) * 8 + (wave >> 1) * 4;
    const int jlane = (lane >> 3) & 3, k0 = (item & 3) * 16 + (wave & 1) * 8, klane = lane & 7;
    constexpr int jspan = 4, kspan = 8, ispan = 2 * kRows, istep = 2;
#endif
    const int ibox = i - ilane;                                          // (wave-uniform: the box's first slab)
    const int key = rec.key, slot = rec.slot;
    if (slot < 0) continue;                                             // pool overflow: reported by the host
    unsigned long long m = rec.mask;
    const int xi = key >> 18, yi = (key >> 9) & 511, zi = key & 511;
    const float xs = unit_shift(xi), ys = unit_shift(yi), zs = unit_shift(zi);
    const float g2 = grid_coord(k0 + klane, zs);
    float2* __restrict__ slab = pool + (size_t)slot * kUnitVox + (size_t)i * (kUnitRes * kUnitRes) + (j0 + jlane) * kUnitRes + k0 + klane;
    float S[kRows], W[kRows], W0[kRows], g0[kRows];                      // g0 per register row and slab of the lane, g1 / g2 per lane
    const float g1 = grid_coord(j0 + jlane, ys);
#pragma unroll
    for (int r = 0; r < kRows; r++) g0[r] = grid_coord(i + r * istep, xs);
    constexpr int row_stride = istep * kUnitRes * kUnitRes;
#pragma unroll
    for (int r = 0; r < kRows; r++) {                                   // loads in flight while the culling preamble computes
      const float2 v = slab[r * row_stride];                      // (loading only the surviving patches, after the culling,
      S[r] = v.x;                                                       //  was measured: no change, the kernel is VALU-bound --
      W[r] = v.y;                                                       //  profiles/r02f_ab_k_integrate_variants.txt)
      W0[r] = v.y;
    }
    // Exact culling: lane f tests frame f of the batch against this wave's patch of 256 voxels; frames that
    // provably cannot update any voxel of the patch leave the mask (er_tsdf_math.h: patch_may_update).
    // The same test also tells which of the remaining frames see the WHOLE patch inside the image and clear of the camera
    // plane (m_in): for those the per-voxel range tests are proven true and the loop below skips them.
    // Third verdict (m_full): the frame updates EVERY voxel of the patch with tsdf = 1 -- proven from the tile minima of the depth
    // under the patch's pixel hull -- so the frame needs no projection, no depth sample and no arithmetic at all: W += 1, and S
    // stays / becomes exactly 1 wherever S == 1 or W == 0 (most of the frustum is such free space).
    unsigned long long m_in, m_full;
    {
      bool keep = ((m >> lane) & 1ull) != 0ull, inside = false, full = false;
#if ER_FRAMES_TRANSPOSED
      // lane f tests frame f: its 16 constants come from the component-major copy behind frames[] (Staging::fxT) -- 64 lanes x 4 consecutive bytes per
      // load where frames[lane] is one 64-byte line per lane
      FrameXform fl;
      if (keep) {
        const float* __restrict__ fT = reinterpret_cast<const float*>(frames + ER_MAX_BATCH) + lane;
#pragma unroll
        for (int q = 0; q < 12; q++) fl.mi[q] = fT[q * ER_MAX_BATCH];
        fl.tx = fT[12 * ER_MAX_BATCH];
        fl.ty = fT[13 * ER_MAX_BATCH];
        fl.tz = fT[14 * ER_MAX_BATCH];
        fl.pad = 0.f;
      }
#else
      const FrameXform& fl = frames[lane];
#endif
      if (keep)
        keep = patch_may_update_box(grid_coord(ibox, xs), grid_coord(ibox + ispan - 1, xs), grid_coord(j0, ys), grid_coord(j0 + jspan - 1, ys),
                                    grid_coord(k0, zs), grid_coord(k0 + kspan - 1, zs), fl, cam, cols, rows,
#if ER_TILE_FRAME_FASTEST
                                    // tiles FRAME-fastest: lane f of this test is frame f, and consecutive frames of a sweep see the box under the
                                    // same tiles -- 64 lanes x 4 consecutive bytes per load instead of 64 lines 1.2 KB apart
                                    tile_max + lane, tiles_x, tiles_y, &inside, tile_lo + lane, &full, kLoShift, lo_tiles_x,
                                    kLoShift < 5 ? tile_lo_fine + lane : (const float*)nullptr, ER_MAX_BATCH);
#else
                                    tile_max + (size_t)lane * tiles_x * tiles_y, tiles_x, tiles_y, &inside,
                                    tile_lo + (size_t)lane * tiles_x * tiles_y, &full, kLoShift, lo_tiles_x,
                                    kLoShift < 5 ? tile_lo_fine + (size_t)lane * lo_tiles : (const float*)nullptr);
#endif
      m = __ballot(keep);
      m_in = __ballot(keep && inside);
      m_full = __ballot(keep && full);
    }
    // Frame loop in two halves: project() computes the pixel under every voxel of the four register rows and issues the depth
    // gathers, finish() does the arithmetic that needs the samples; the loop below overlaps the two halves of consecutive frames.
    auto project = [&](int f, float (&dp)[kRows]) {
      const FrameXform fx = frames[f];
      const float* __restrict__ sc = scaled + (size_t)f * (pixels + kScaledPad);
      unsigned pix[kRows];
      if ((m_in >> f) & 1ull) {                                          // wave-uniform
#pragma unroll
        for (int r = 0; r < kRows; r++) pix[r] = voxel_project_inside(g0[r], g1, g2, fx, cam, cols, rows);
      } else
      {
#pragma unroll
        for (int r = 0; r < kRows; r++) {
          unsigned pixel;
          const bool ok = voxel_project(g0[r], g1, g2, fx, cam, cols, rows, pixel);
          pix[r] = ok ? pixel : (unsigned)pixels;                        // the frame's zero pad: dp = 0 fails ":82 dp > 0.001" like the reference's early out
        }
      }
      // kRows UNCONDITIONAL gathers in straight-line code after the branches, nothing that depends on them here: the wait in
      // finish() is then "all but the newest kRows loads" on every path (predicated loads or loads inside the branches make the
      // count path-dependent and the compiler falls back to waiting for everything)
#pragma unroll
      for (int r = 0; r < kRows; r++) dp[r] = sc[pix[r]];
    };
    auto finish = [&](int f, const float (&dp)[kRows]) {
      const FrameXform& fx = frames[f];                                  // (only the camera centre: three scalar loads)
      float d2[kRows];
#pragma unroll
      for (int r = 0; r < kRows; r++) d2[r] = voxel_dist2(g0[r], g1, g2, fx);
      if (kSure) {
        // Sure path (er_tsdf_math.h: voxel_classify): if every lane of the four rows is provably in free space (tsdf = 1) or
        // provably behind the surface (no update) and every free lane holds S == 1 or W == 0, the whole update of this frame is
        // "W += 1, S = 1" on the free lanes -- no square root, no band quotient, no division.  78 % of the (patch, frame)
        // visits of the golden scene; one wave-uniform branch per frame.
        bool fre[kRows], need = false;
#pragma unroll
        for (int r = 0; r < kRows; r++) {
          bool behind;
          voxel_classify(dp[r], d2[r], fre[r], behind);
          need = need | !(fre[r] | behind) | (fre[r] & !voxel_free_trivial(S[r], W[r]));
        }
        if (__ballot(need) == 0ull) {
#pragma unroll
          for (int r = 0; r < kRows; r++) {
            S[r] = fre[r] ? 1.0f : S[r];
            W[r] = fre[r] ? W[r] + 1.0f : W[r];
          }
          return;
        }
      }
#pragma unroll
      for (int r = 0; r < kRows; r++) {
        const bool upd = voxel_finish_d2(S[r], W[r], dp[r], d2[r]);
        (void)upd;
      }
    };
    // Software pipeline over the frames that need a projection: the projection and the four depth gathers of the NEXT such frame are issued before the
    // current frame's samples are used (a wave used to sit on its gathers once per frame: 3100 ticks per visit for ~730 issue cycles).  Runs of full
    // frames need no samples and are applied where they fall in the ascending order, so every voxel still sees its frames one by one in frame order.
    // Two stages per trip with alternating sample registers (a rotating copy would have to wait for the data it copies); the last frame is finished
    // after the loop.  +4 % for the job together with three instead of four persistent workgroups per CU (profiles/r05n_ab_frame_pipeline.txt).
    {
      unsigned long long mn = m & ~m_full, mf = m & m_full;
      auto apply_full = [&](unsigned long long run) {
        const int n = __popcll(run);
        bool nontrivial = false;
#pragma unroll
        for (int r = 0; r < kRows; r++) nontrivial = nontrivial | !(voxel_free_trivial(S[r], W[r]) & (W[r] < 8388608.0f));
        if (__ballot(nontrivial) == 0ull) {                              // (S W + 1) / (W + 1) == 1 exactly, W + n exact below 2^24
#pragma unroll
          for (int r = 0; r < kRows; r++) {
            S[r] = 1.0f;
            W[r] = W[r] + (float)n;
          }
        } else {                                                         // a voxel that was inside the truncation band before: the n divisions, in order
          for (int q = 0; q < n; q++) {
#pragma unroll
            for (int r = 0; r < kRows; r++) {
              S[r] = div_inrange(S[r] * W[r] + 1.0f, W[r] + 1.0f);
              W[r] = W[r] + 1.0f;
            }
          }
        }
      };
      auto runs_before = [&](int f) {
        const unsigned long long run = f < 64 ? (mf & ((1ull << f) - 1ull)) : mf;   // the full frames before the next projected one
        if (run) {                                                       // wave-uniform
          mf &= ~run;
          apply_full(run);
        }
      };
      if (mn) {
        float dpa[kRows], dpb[kRows];
        int pf = __builtin_ctzll(mn);
        mn &= mn - 1;
        project(pf, dpa);
        bool last_in_b = false;
        for (;;) {                                                       // two stages per trip: the sample registers alternate; the last frame is finished after the loop
          runs_before(pf);
          if (mn == 0ull) break;                                         // (pf's samples are in dpa)
          int nf = __builtin_ctzll(mn);
          mn &= mn - 1;
          project(nf, dpb);
          finish(pf, dpa);
          pf = nf;
          runs_before(pf);
          if (mn == 0ull) {                                              // (pf's samples are in dpb)
            last_in_b = true;
            break;
          }
          nf = __builtin_ctzll(mn);
          mn &= mn - 1;
          project(nf, dpa);
          finish(pf, dpb);
          pf = nf;
        }
        float dpl[kRows];
#pragma unroll
        for (int r = 0; r < kRows; r++) dpl[r] = last_in_b ? dpb[r] : dpa[r];
        finish(pf, dpl);                                                 // the last projected frame: nothing left to prefetch
      }
      runs_before(64);                                                   // the full frames after the last projected one
    }
#pragma unroll
    for (int r = 0; r < kRows; r++)
      if (W[r] != W0[r]) slab[r * row_stride] = make_float2(S[r], W[r]);
  }
}
template __global__ void k_integrate<true>(float2* __restrict__, const PlanRec* __restrict__, Plan* __restrict__, const FrameXform* __restrict__, const float* __restrict__, const float* __restrict__, const float* __restrict__, const float* __restrict__, int, int, Camera, int, int);
template __global__ void k_integrate<false>(float2* __restrict__, const PlanRec* __restrict__, Plan* __restrict__, const FrameXform* __restrict__, const float* __restrict__, const float* __restrict__, const float* __restrict__, const float* __restrict__, int, int, Camera, int, int);
}  // namespace er_tsdf_k
#else
namespace er_tsdf_k {
extern template __global__ void k_integrate<true>(float2* __restrict__, const PlanRec* __restrict__, Plan* __restrict__, const FrameXform* __restrict__, const float* __restrict__, const float* __restrict__, const float* __restrict__, const float* __restrict__, int, int, Camera, int, int);
extern template __global__ void k_integrate<false>(float2* __restrict__, const PlanRec* __restrict__, Plan* __restrict__, const FrameXform* __restrict__, const float* __restrict__, const float* __restrict__, const float* __restrict__, const float* __restrict__, int, int, Camera, int, int);
}  // namespace er_tsdf_k
#endif  // ER_TSDF_TU == 2
#if ER_TSDF_TU == 0
namespace {

// Clears the frame masks of the batch list and accounts unit visits (sum of popcounts).
__global__ void k_reset(const int* __restrict__ batch, int* __restrict__ nbatch, unsigned long long* __restrict__ ht_mask,
                        unsigned long long* __restrict__ stats) {
  const int n = *nbatch;
  unsigned long long visits = 0;
  for (int t = threadIdx.x; t < n; t += blockDim.x) {
    const int e = batch[t];
    visits += (unsigned long long)__popcll(ht_mask[e]);
    ht_mask[e] = 0ull;
  }
  if (visits) atomicAdd(&stats[0], visits);
  __syncthreads();
  if (threadIdx.x == 0) *nbatch = 0;
}

// ------------------------------------------------------------------------------------------------
// Sum of weight_ (= number of voxel updates, TSDFVolume.cpp:90,94); wave shuffle -> one atomic per block.
__global__ void k_sum_weight(const float2* __restrict__ pool, long n_vox, double* __restrict__ out) {
  double s = 0.0;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n_vox; t += (long)gridDim.x * blockDim.x)
    s += (double)pool[t].y;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  __shared__ double part[kBlock / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double b = 0.0;
    for (int w = 0; w < kBlock / 64; w++) b += part[w];
    atomicAdd(out, b);
  }
}

// SaveWorld's filter (TSDFVolume.cpp:118).  One wave per (unit, i-slab); pass 0 counts, pass 1 writes
// the points in i,j,k order at the slab's offset (stable compaction by ballot prefix).
__device__ __forceinline__ bool world_keep(float2 v) { return v.y != 0.0f && v.x < 0.98f && v.x >= -0.98f; }

__global__ __launch_bounds__(64) void k_world(const float2* __restrict__ pool, const int* __restrict__ slots,
                                              const int* __restrict__ keys, long* __restrict__ slab_count,
                                              const long* __restrict__ slab_offset, float4* __restrict__ out, int pass) {
  const int rank = blockIdx.x >> 6;          // unit in ascending key order
  const int i = blockIdx.x & 63;
  const int lane = threadIdx.x;
  const float2* slab = pool + (size_t)slots[rank] * kUnitVox + (size_t)i * 4096;
  const int key = keys[rank];
  const int xi = key >> 18, yi = (key >> 9) & 511, zi = key & 511;
  long base = pass ? slab_offset[blockIdx.x] : 0;
  long total = 0;
  for (int j = 0; j < 64; j++) {
    const float2 v = slab[j * 64 + lane];
    const bool keep = world_keep(v);
    const unsigned long long b = __ballot(keep);
    if (pass && keep) {
      const long o = base + total + __popcll(b & ((1ull << lane) - 1ull));
      out[o] = make_float4((float)(i + (xi - 256) * 64), (float)(j + (yi - 256) * 64), (float)(lane + (zi - 256) * 64), v.x);
    }
    total += __popcll(b);
  }
  if (!pass && lane == 0) slab_count[blockIdx.x] = total;
}

// Zero-crossing extraction on the resident volume (what the out-of-repo kinfu "mesh_output" step does with
// world.pcd, done where the volume lives).  For every observed voxel (weight != 0) and each of its +x, +y, +z neighbours --
// inside the unit or in the adjacent unit, found through the hash map -- that is observed too: if the two sdf values have
// strictly opposite signs, the surface crosses that lattice edge at t = F / (F - Fn) and the point
//     p = voxel position + t * voxel size along the axis            (float32; position = (float)(global index * 3/512))
// is emitted (kinfu's extractCloud rule).  Order: units by ascending key, voxels in i,j,k order, axes x,y,z -- a stable
// ballot-prefix compaction in two passes like k_world, so the list is reproducible and a CPU restatement can match it
// element for element (tests/test_tsdf_gpu.py).
__device__ __forceinline__ int ht_lookup_slot(const int* __restrict__ ht_key, const int* __restrict__ ht_slot, int cap_mask, int shift, int key) {
  unsigned h = hash_unit_key(key, shift);
  for (int probe = 0; probe <= cap_mask; ++probe) {
    const int e = (int)((h + (unsigned)probe) & (unsigned)cap_mask);
    const int k = ht_key[e];
    if (k == key) return ht_slot[e];
    if (k == kEmptyKey) return -1;
  }
  return -1;
}

__device__ __forceinline__ bool crosses(float2 a, float2 b) {
  return a.y != 0.0f && b.y != 0.0f && ((a.x > 0.0f && b.x < 0.0f) || (a.x < 0.0f && b.x > 0.0f));
}

__global__ __launch_bounds__(64) void k_surface(const float2* __restrict__ pool, const int* __restrict__ slots, const int* __restrict__ keys,
                                                const int* __restrict__ ht_key, const int* __restrict__ ht_slot, int cap_mask, int shift,
                                                long* __restrict__ slab_count, const long* __restrict__ slab_offset,
                                                float4* __restrict__ out, int pass) {
  const int rank = blockIdx.x >> 6;          // unit in ascending key order
  const int i = blockIdx.x & 63;
  const int lane = threadIdx.x;              // = k
  const int key = keys[rank];
  const int xi = key >> 18, yi = (key >> 9) & 511, zi = key & 511;
  const float2* unit = pool + (size_t)slots[rank] * kUnitVox;
  const float2* slab = unit + (size_t)i * 4096;
  // neighbours that live in adjacent units (wave-uniform lookups; -1 = that unit does not exist)
  const int sx = (i == 63 && xi < 511) ? ht_lookup_slot(ht_key, ht_slot, cap_mask, shift, key + 512 * 512) : -1;
  const int sy = yi < 511 ? ht_lookup_slot(ht_key, ht_slot, cap_mask, shift, key + 512) : -1;
  const int sz = zi < 511 ? ht_lookup_slot(ht_key, ht_slot, cap_mask, shift, key + 1) : -1;
  const float2 none = make_float2(0.0f, 0.0f);
  const float2* slab_x = i < 63 ? slab + 4096 : (sx >= 0 ? pool + (size_t)sx * kUnitVox : nullptr);              // i + 1 (slab 0 of the next unit)
  const float2* unit_y = sy >= 0 ? pool + (size_t)sy * kUnitVox + (size_t)i * 4096 : nullptr;                      // j + 1 == 64: row 0 there
  const float2* unit_z = sz >= 0 ? pool + (size_t)sz * kUnitVox + (size_t)i * 4096 : nullptr;                      // k + 1 == 64: voxel 0 there
  const float ulf = (float)kUnitLength;
  const float gx = (float)((double)(i + (xi - 256) * 64) * kUnitLength);
  const float gz = (float)((double)(lane + (zi - 256) * 64) * kUnitLength);
  const unsigned long long lt = (1ull << lane) - 1ull;
  long base = pass ? slab_offset[blockIdx.x] : 0;
  long total = 0;
  for (int j = 0; j < 64; j++) {
    const float2 v = slab[j * 64 + lane];
    const float2 nx = slab_x ? slab_x[j * 64 + lane] : none;
    const float2 ny = j < 63 ? slab[(j + 1) * 64 + lane] : (unit_y ? unit_y[lane] : none);
    float2 nz;
    nz.x = __shfl_down(v.x, 1);
    nz.y = __shfl_down(v.y, 1);
    if (lane == 63) nz = unit_z ? unit_z[j * 64] : none;
    const bool cx = crosses(v, nx), cy = crosses(v, ny), cz = crosses(v, nz);
    const unsigned long long bx = __ballot(cx), by = __ballot(cy), bz = __ballot(cz);
    if (pass) {
      long o = base + total + __popcll(bx & lt) + __popcll(by & lt) + __popcll(bz & lt);
      const float gy = (float)((double)(j + (yi - 256) * 64) * kUnitLength);
      if (cx) out[o++] = make_float4(gx + (v.x / (v.x - nx.x)) * ulf, gy, gz, 0.0f);
      if (cy) out[o++] = make_float4(gx, gy + (v.x / (v.x - ny.x)) * ulf, gz, 1.0f);
      if (cz) out[o++] = make_float4(gx, gy, gz + (v.x / (v.x - nz.x)) * ulf, 2.0f);
    }
    total += __popcll(bx) + __popcll(by) + __popcll(bz);
  }
  if (!pass && lane == 0) slab_count[blockIdx.x] = total;
}


// Marching cubes on the resident volume (the triangle connectivity the out-of-repo kinfu "mesh_output" step builds
// from world.pcd, done where the volume lives).  Cell (i, j, k) of a unit = the eight voxels (i..i+1, j..j+1, k..k+1) -- the last
// layer of cells reaches into the adjacent units (+x, +y, +z and their combinations, found through the hash map).  A cell
// yields triangles only if all eight voxels are observed (weight != 0, kinfu's rule); corner c is inside iff sdf < 0; the case
// table is generated on the host (er_mc_table.h) and staged in LDS.  A vertex on the lattice edge from the lower voxel L to the
// upper voxel H lies at  pos(L) + (F_L / (F_L - F_H)) * voxel size  along the edge's axis (float32; pos = (float)(global index *
// 3/512)) -- evaluated from the edge's LOWER end whichever cell asks, so the cells that share the edge produce the same bits and
// the triangle soup is watertight by vertex equality.  Order: units by ascending key, cells in i, j, k order, triangles in table
// order; two passes (count, then write at the slab's offset: a stable ballot-prefix compaction) like k_world / k_surface.
__global__ __launch_bounds__(64) void k_mesh(const float2* __restrict__ pool, const int* __restrict__ slots, const int* __restrict__ keys,
                                             const int* __restrict__ ht_key, const int* __restrict__ ht_slot, int cap_mask, int shift,
                                             const unsigned char* __restrict__ table, long* __restrict__ slab_count,
                                             const long* __restrict__ slab_offset, float* __restrict__ out, int pass) {
  __shared__ unsigned char s_tab[256 * 16];
  for (int t = threadIdx.x; t < 256 * 16 / 4; t += 64) reinterpret_cast<unsigned*>(s_tab)[t] = reinterpret_cast<const unsigned*>(table)[t];
  __syncthreads();
  const int rank = blockIdx.x >> 6;          // unit in ascending key order
  const int i = blockIdx.x & 63;
  const int lane = threadIdx.x;              // = k
  const int key = keys[rank];
  const int xi = key >> 18, yi = (key >> 9) & 511, zi = key & 511;
  // the (up to) eight units a slab of cells can touch: [dx][dy][dz]; -1 = that unit does not exist (its voxels count as unobserved)
  int us[2][2][2];
  for (int dx = 0; dx < 2; dx++)
    for (int dy = 0; dy < 2; dy++)
      for (int dz = 0; dz < 2; dz++) {
        const bool need = (dx == 0 || i == 63);
        const bool ok = xi + dx < 512 && yi + dy < 512 && zi + dz < 512;
        us[dx][dy][dz] = (dx | dy | dz) == 0 ? slots[rank]
                         : (need && ok ? ht_lookup_slot(ht_key, ht_slot, cap_mask, shift, key + dx * 512 * 512 + dy * 512 + dz) : -1);
      }
  const int ia[2] = {i, i == 63 ? 0 : i + 1}, ux[2] = {0, i == 63 ? 1 : 0};      // slab index and unit offset of the two i layers
  const float2 none = make_float2(0.0f, 0.0f);
  const float ulf = (float)kUnitLength;
  const float gx = (float)((double)(i + (xi - 256) * 64) * kUnitLength);
  const float gz = (float)((double)(lane + (zi - 256) * 64) * kUnitLength);
  const unsigned long long lt = (1ull << lane) - 1ull;
  long base = pass ? slab_offset[blockIdx.x] : 0;
  long total = 0;
  for (int j = 0; j < 64; j++) {
    // the eight corners of this lane's cell: f[a][b][c] = voxel (i + a, j + b, k + c)
    float2 f[2][2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int jb = (j + b) & 63, uy = (j + b) >> 6;
        const int s0 = us[ux[a]][uy][0], s1 = us[ux[a]][uy][1];
        const size_t ro = (size_t)ia[a] * 4096 + (size_t)jb * 64;
        const float2 v = s0 >= 0 ? pool[(size_t)s0 * kUnitVox + ro + lane] : none;
        f[a][b][0] = v;
        float2 w;
        w.x = __shfl_down(v.x, 1);
        w.y = __shfl_down(v.y, 1);
        if (lane == 63) w = s1 >= 0 ? pool[(size_t)s1 * kUnitVox + ro] : none;
        f[a][b][1] = w;
      }
    bool valid = true;
    int cs = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const float2 v = f[c & 1][(c >> 1) & 1][c >> 2];
      valid = valid && v.y != 0.0f;
      cs |= (v.x < 0.0f ? 1 : 0) << c;
    }
    const unsigned char* __restrict__ row = s_tab + cs * 16;
    int nt = 0;
    if (valid)
      while (nt < 5 && row[3 * nt] != 255) nt++;
    // wave-level exclusive prefix of the triangle counts (k order)
    int incl = nt;
    for (int sft = 1; sft < 64; sft <<= 1) {
      const int t = __shfl_up(incl, sft);
      if (lane >= sft) incl += t;
    }
    const int wave_total = __shfl(incl, 63);
    if (pass && nt > 0) {
      const float gy = (float)((double)(j + (yi - 256) * 64) * kUnitLength);
      float* __restrict__ o = out + (size_t)(base + total + (incl - nt)) * 9;
      for (int t = 0; t < 3 * nt; t++) {
        const int e = row[t];
        const int axis = e >> 2, u = e & 1, v = (e >> 1) & 1;
        // lower corner of the edge (coordinate 0 along its axis) and the corner one step up the axis; the eight corner values sit
        // in registers, so they are picked with select chains, not with a runtime index (that would send them through scratch)
        const int a0 = axis == 0 ? 0 : u, b0 = axis == 1 ? 0 : (axis == 0 ? u : v), c0 = axis == 2 ? 0 : v;
        const int cl = a0 | b0 << 1 | c0 << 2, ch = cl | (1 << axis);
        float2 lo = none, hi = none;
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const float2 fv = f[c & 1][(c >> 1) & 1][c >> 2];
          lo = c == cl ? fv : lo;
          hi = c == ch ? fv : hi;
        }
        const float tt = lo.x / (lo.x - hi.x);
        // lower end of the edge: lattice position of voxel (i + a0, j + b0, k + c0)
        float px = a0 ? (float)((double)(i + 1 + (xi - 256) * 64) * kUnitLength) : gx;
        float py = b0 ? (float)((double)(j + 1 + (yi - 256) * 64) * kUnitLength) : gy;
        float pz = c0 ? (float)((double)(lane + 1 + (zi - 256) * 64) * kUnitLength) : gz;
        if (axis == 0) px = px + tt * ulf;
        if (axis == 1) py = py + tt * ulf;
        if (axis == 2) pz = pz + tt * ulf;
        o[3 * t] = px;
        o[3 * t + 1] = py;
        o[3 * t + 2] = pz;
      }
    }
    total += wave_total;
  }
  (void)lt;
  if (!pass && lane == 0) slab_count[blockIdx.x] = total;
}

// Multi-GPU frame split: planes [key][0] = sdf*weight, [key][1] = weight -- what a sum over ranks may add (units several ranks touched);
// raw != 0: [key][0] = sdf, [key][1] = weight, the unit bit for bit (units only one rank touched travel like this, round 5).
__global__ void k_export_weighted(const float2* __restrict__ pool, const int* __restrict__ slots, float* __restrict__ buf, int raw) {
  const int q = blockIdx.y;
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  const int slot = slots[q];
  float sw = 0.0f, w = 0.0f;
  if (slot >= 0) {
    const float2 v = pool[(size_t)slot * kUnitVox + l];
    sw = raw ? v.x : v.x * v.y;
    w = v.y;
  }
  buf[((size_t)q * 2 + 0) * kUnitVox + l] = sw;
  buf[((size_t)q * 2 + 1) * kUnitVox + l] = w;
}

__global__ void k_import_weighted(float2* __restrict__ pool, const int* __restrict__ slots, const float* __restrict__ buf, int raw) {
  const int q = blockIdx.y;
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  const int slot = slots[q];
  if (slot < 0) return;
  const float sw = buf[((size_t)q * 2 + 0) * kUnitVox + l];
  const float w = buf[((size_t)q * 2 + 1) * kUnitVox + l];
  pool[(size_t)slot * kUnitVox + l] = make_float2(raw ? sw : (w > 0.0f ? sw / w : 0.0f), w);
}

// ---- band records (round 6: the owner merge of the frame split, csrc/er_merge_protocol.h) ----------------------------------------------------
// A unit as its OBSERVED voxels only (weight != 0; measured on configs[3]: 0.28 of a touched unit), and of those the sdf only where it is not exactly 1
// -- free space in front of a surface: every frame wrote tsdf = 1 there, so the running mean is 1.0f to the bit; 81 % of the observed voxels --, and the
// weight, a frame count, as 16 bits when every weight of the unit fits (flag bit 0 otherwise: float32 weights).  32-bit words:
//   [0] flags  [1] observed voxels  [2] band voxels (observed, sdf != 1)  [3] 0
//   [4, 132)          exclusive prefix of the observed-voxel counts of the unit's 128 chunks of 2048 voxels (a chunk = one wave's share)
//   [132, 260)        ... of the band-voxel counts
//   [260, 8452)       observed bitmap, bit (l & 63) of the 64-bit word l >> 6 <-> voxel l (k fastest, like the pool)
//   [8452, 16644)     sdf-is-one bitmap (a subset of the observed one)
//   then              the weights of the observed voxels in voxel order (uint16, or float32 with flag bit 0), padded to an even number of words,
//   then              the sdf_ of the band voxels in voxel order (float32), padded to an even number of words.
// A never-updated voxel is (+0, 0) in the pool (TSDFVolumeUnit.cpp:4-21 zero-fills, TSDFVolume.cpp:93-94 writes both), so a record restores a unit bit for bit.
constexpr int kBandChunk = 2048;
constexpr int kBandChunks = kUnitVox / kBandChunk;          // 128
constexpr int kBandBitmapWords = kUnitVox / 32;             // 8192
constexpr int kBandObsPrefix = 4, kBandBandPrefix = kBandObsPrefix + kBandChunks, kBandObsBits = kBandBandPrefix + kBandChunks,
              kBandOneBits = kBandObsBits + kBandBitmapWords, kBandHeader = kBandOneBits + kBandBitmapWords;   // 16 644 words before the values
constexpr int kBandMaxSrc = 16;
constexpr uint32_t kOneBits = 0x3f800000u;

__host__ __device__ inline long band_weight_words(int obs, int wide) { return wide ? (long)((obs + 1) & ~1) : 2L * ((obs + 3) / 4); }
__host__ __device__ inline long band_record_words(int obs, int band, int wide) { return (long)kBandHeader + band_weight_words(obs, wide) + (long)((band + 1) & ~1); }

// counts[q][0..127] observed, [128..255] band voxels per chunk; wide[q] |= 1 if a weight does not fit 16 bits
__global__ __launch_bounds__(256) void k_band_count(const float2* __restrict__ pool, const int* __restrict__ slots, int* __restrict__ counts, int* __restrict__ wide) {
  const int q = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int chunk = blockIdx.x * 4 + wave;
  const int slot = slots[q];
  int n = 0, nb = 0, w = 0;
  if (slot >= 0) {
    const float2* __restrict__ u = pool + (size_t)slot * kUnitVox + (size_t)chunk * kBandChunk;
#pragma unroll 8
    for (int it = 0; it < kBandChunk / 64; it++) {
      const float2 v = u[it * 64 + lane];
      const bool on = v.y != 0.0f;
      n += __popcll(__ballot(on));
      nb += __popcll(__ballot(on && __float_as_uint(v.x) != kOneBits));
      w |= (on && !(v.y >= 1.0f && v.y <= 65535.0f && v.y == floorf(v.y))) ? 1 : 0;
    }
  }
  if (lane == 0) {
    counts[q * 2 * kBandChunks + chunk] = n;
    counts[q * 2 * kBandChunks + kBandChunks + chunk] = nb;
  }
  if (__any(w) && lane == 0) atomicOr(&wide[q], 1);
}

__global__ __launch_bounds__(256) void k_band_pack(const float2* __restrict__ pool, const int* __restrict__ slots, const int* __restrict__ counts,
                                                   const int* __restrict__ wide, const long* __restrict__ rec_off, uint32_t* __restrict__ out) {
  const int q = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int chunk = blockIdx.x * 4 + wave;
  const int slot = slots[q];
  uint32_t* __restrict__ rec = out + rec_off[q];
  const int* __restrict__ co = counts + q * 2 * kBandChunks;
  int before = (lane < chunk ? co[lane] : 0) + (lane + 64 < chunk ? co[64 + lane] : 0);
  int before_b = (lane < chunk ? co[kBandChunks + lane] : 0) + (lane + 64 < chunk ? co[kBandChunks + 64 + lane] : 0);
  int total = co[lane] + co[64 + lane], total_b = co[kBandChunks + lane] + co[kBandChunks + 64 + lane];
  for (int o = 32; o > 0; o >>= 1) {
    before += __shfl_xor(before, o);
    before_b += __shfl_xor(before_b, o);
    total += __shfl_xor(total, o);
    total_b += __shfl_xor(total_b, o);
  }
  const int is_wide = wide[q] & 1;
  if (lane == 0) {
    rec[kBandObsPrefix + chunk] = (uint32_t)before;
    rec[kBandBandPrefix + chunk] = (uint32_t)before_b;
    if (chunk == 0) {
      rec[0] = (uint32_t)is_wide;
      rec[1] = (uint32_t)total;
      rec[2] = (uint32_t)total_b;
      rec[3] = 0u;
    }
  }
  uint32_t* __restrict__ wts = rec + kBandHeader;
  float* __restrict__ sdf = reinterpret_cast<float*>(rec + kBandHeader + band_weight_words(total, is_wide));
  unsigned long long* __restrict__ bits = reinterpret_cast<unsigned long long*>(rec + kBandObsBits) + (size_t)chunk * (kBandChunk / 64);
  unsigned long long* __restrict__ ones = reinterpret_cast<unsigned long long*>(rec + kBandOneBits) + (size_t)chunk * (kBandChunk / 64);
  const float2* __restrict__ u = pool + (size_t)(slot < 0 ? 0 : slot) * kUnitVox + (size_t)chunk * kBandChunk;
  const unsigned long long below = (1ull << lane) - 1ull;
  int off = before, off_b = before_b;
#pragma unroll 4
  for (int it = 0; it < kBandChunk / 64; it++) {
    const float2 v = slot < 0 ? make_float2(0.f, 0.f) : u[it * 64 + lane];
    const bool on = v.y != 0.0f, one = on && __float_as_uint(v.x) == kOneBits;
    const unsigned long long b = __ballot(on), b1 = __ballot(one);
    if (lane == 0) {
      bits[it] = b;
      ones[it] = b1;
    }
    if (on) {
      const int at = off + __popcll(b & below);
      if (is_wide) reinterpret_cast<float*>(wts)[at] = v.y;
      else reinterpret_cast<unsigned short*>(wts)[at] = (unsigned short)v.y;
      if (!one) sdf[off_b + __popcll((b & ~b1) & below)] = v.x;
    }
    off += __popcll(b);
    off_b += __popcll(b & ~b1);
  }
}

struct BandItem {
  int slot, nsrc, self_pos, pad;
  const uint32_t* rec[kBandMaxSrc];
};

// voxel (chunk, it, lane) of a record: {sdf, weight} or (0, 0); o / ob = the wave's running offsets into the record's weights / band values
__device__ __forceinline__ float2 band_fetch(const uint32_t* __restrict__ rec, int chunk, int it, int lane, unsigned long long below, int& o, int& ob) {
  const unsigned long long b = reinterpret_cast<const unsigned long long*>(rec + kBandObsBits)[(size_t)chunk * (kBandChunk / 64) + it];
  const unsigned long long b1 = reinterpret_cast<const unsigned long long*>(rec + kBandOneBits)[(size_t)chunk * (kBandChunk / 64) + it];
  float2 v = make_float2(0.0f, 0.0f);
  if ((b >> lane) & 1ull) {
    const int is_wide = (int)(rec[0] & 1u), total = (int)rec[1];
    const uint32_t* __restrict__ wts = rec + kBandHeader;
    const int at = o + __popcll(b & below);
    v.y = is_wide ? reinterpret_cast<const float*>(wts)[at] : (float)reinterpret_cast<const unsigned short*>(wts)[at];
    v.x = ((b1 >> lane) & 1ull) ? 1.0f : reinterpret_cast<const float*>(rec + kBandHeader + band_weight_words(total, is_wide))[ob + __popcll((b & ~b1) & below)];
  }
  o += __popcll(b);
  ob += __popcll(b & ~b1);
  return v;
}

// The owner's sum of one unit: its own voxels and the records of the other touchers IN RANK ORDER (self_pos = records that come before its own):
//   SW = sum_r fl(sdf_r * w_r), W = sum_r w_r, sdf = SW / W   -- TSDFVolume.cpp:93-94 as a sum, what k_export_weighted + a rank-ordered reduction +
// k_import_weighted compute, with the order fixed by the key sets (this translation unit is compiled with -ffp-contract=off: product, then sum).
__global__ __launch_bounds__(256) void k_band_merge(float2* __restrict__ pool, const BandItem* __restrict__ items) {
  const BandItem& item = items[blockIdx.y];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int chunk = blockIdx.x * 4 + wave;
  __shared__ int off[4][kBandMaxSrc][2];
  const int nsrc = item.nsrc, self_pos = item.self_pos;
  if (lane < nsrc) {
    off[wave][lane][0] = (int)item.rec[lane][kBandObsPrefix + chunk];
    off[wave][lane][1] = (int)item.rec[lane][kBandBandPrefix + chunk];
  }
  float2* __restrict__ u = pool + (size_t)item.slot * kUnitVox + (size_t)chunk * kBandChunk;
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int it = 0; it < kBandChunk / 64; it++) {
    float sw = 0.0f, w = 0.0f;
    const float2 own = u[it * 64 + lane];
    for (int s = 0; s <= nsrc; s++) {
      if (s == self_pos) {
        sw += own.x * own.y;
        w += own.y;
      }
      if (s == nsrc) break;
      int o = off[wave][s][0], ob = off[wave][s][1];
      const float2 v = band_fetch(item.rec[s], chunk, it, lane, below, o, ob);
      sw += v.x * v.y;                                           // (an unobserved voxel adds +0: the same bits as skipping it)
      w += v.y;
      if (lane == 0) {
        off[wave][s][0] = o;
        off[wave][s][1] = ob;
      }
    }
    u[it * 64 + lane] = w > 0.0f ? make_float2(sw / w, w) : make_float2(0.0f, 0.0f);
  }
}

// record -> unit, bit for bit (every voxel is written: an unobserved one becomes (+0, 0))
__global__ __launch_bounds__(256) void k_band_import(float2* __restrict__ pool, const int* __restrict__ slots, const uint32_t* const* __restrict__ recs) {
  const int q = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int chunk = blockIdx.x * 4 + wave;
  const int slot = slots[q];
  if (slot < 0) return;
  const uint32_t* __restrict__ rec = recs[q];
  float2* __restrict__ u = pool + (size_t)slot * kUnitVox + (size_t)chunk * kBandChunk;
  int o = (int)rec[kBandObsPrefix + chunk], ob = (int)rec[kBandBandPrefix + chunk];
  const unsigned long long below = (1ull << lane) - 1ull;
  for (int it = 0; it < kBandChunk / 64; it++) u[it * 64 + lane] = band_fetch(rec, chunk, it, lane, below, o, ob);
}

__global__ __launch_bounds__(256) void k_zero_units(float2* __restrict__ pool, const int* __restrict__ slots) {
  const int slot = slots[blockIdx.y];
  if (slot < 0) return;
  float4* __restrict__ u = reinterpret_cast<float4*>(pool + (size_t)slot * kUnitVox);
  u[blockIdx.x * 256 + threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Host-driven unit allocation (import of units this GPU never touched).
__global__ void k_ensure_units(const int* __restrict__ keys, int n, int* __restrict__ ht_key, int* __restrict__ ht_slot,
                               int cap_mask, int hash_shift, int* __restrict__ unit_key, int max_units,
                               int* __restrict__ counters, int* __restrict__ slots_out, int allocate) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  const int key = keys[t];
  int slot = -1;
  if (allocate) {
    const int e = ht_find_or_insert(ht_key, cap_mask, hash_shift, key);   // keys are unique: no race on the slot
    if (e < 0) {
      atomicOr(&counters[C_TABLE_FULL], 1);
    } else {
      slot = ht_slot[e];
      if (slot < 0) {
        const int s = atomicAdd(&counters[C_NUNITS], 1);
        if (s < max_units) {
          ht_slot[e] = s;
          unit_key[s] = key;
          slot = s;
        } else {
          atomicOr(&counters[C_POOL_OVERFLOW], 1);
        }
      }
    }
  } else {
    unsigned h = hash_unit_key(key, hash_shift);
    for (int probe = 0; probe <= cap_mask; ++probe) {
      const int e = (int)((h + (unsigned)probe) & (unsigned)cap_mask);
      const int k = ht_key[e];
      if (k == key) { slot = ht_slot[e]; break; }
      if (k == kEmptyKey) break;
    }
  }
  slots_out[t] = slot;
}

}  // namespace

// ================================================================================================
struct er_tsdf_s {
  int device = 0, cols = 0, rows = 0, pixels = 0, max_units = 0;
  er::Camera cam{};
  er::CameraInv cami{};
  hipStream_t own_stream = nullptr, stream = nullptr;   // `stream` carries k_plan/k_integrate/k_reset and every other call
  hipStream_t aux_stream[kAux] = {};                      // pre-passes (reproject, prepare) of the NEXT TWO batches run here, overlapped
  hipStream_t copy_stream = nullptr;                      // host depth -> depth_stage[slot], overlapped with all of the above; created on
                                                          // first use (HIP multiplexes streams over 4 hardware queues by default, see er_tsdf_create)
  hipEvent_t copy_done[kDepth] = {};
  hipEvent_t consts_done[kDepth] = {};                    // the per-batch constants of slot q have left the pinned block
  int n_cu = 256;
  int shard_rank = 0, shard_world = 1;                    // unit-shard mode (er_tsdf_set_unit_shard)
  // device memory
  float2* pool = nullptr;
  int *ht_key = nullptr, *ht_slot = nullptr, *unit_key = nullptr, *counters = nullptr;
  unsigned long long* stats = nullptr;
  // triple-buffered batch state (three batches in flight: pre-passes of n+1 and n+2 overlap k_integrate of n)
  long batch_no = 0;                                // batch b uses slot b mod kDepth and pre-pass stream b mod kAux
  bool used[kDepth] = {};
  int* batch[kDepth] = {};
  unsigned long long* ht_mask[kDepth] = {};
  float *scaled[kDepth] = {}, *tile_max[kDepth] = {}, *tile_lo[kDepth] = {}, *tile_lo_fine[kDepth] = {};
  er::FrameXform* frames[kDepth] = {};              // = &dstage[q]->fx
  void* dstage[kDepth] = {};                        // device twin of the pinned per-batch constants (struct Staging)
  hipEvent_t pre_done[kDepth] = {}, int_done[kDepth] = {};
  void* pinned[kDepth] = {};                              // host staging of the per-batch constants
  int ht_cap = 0, ht_shift = 0;
  float *lambda = nullptr, *ctr = nullptr;
  // The caller's lattices, double-buffered by call parity on the host (page-locked staging) AND on the device, so that the
  // upload of call c (copy stream) never waits for the pre-passes of call c-1 that still read the other buffer.
  float* ctr_pinned[2] = {nullptr, nullptr};
  float* ctr_dev[2] = {nullptr, nullptr};
  size_t ctr_pinned_cap[2] = {0, 0}, ctr_dev_cap[2] = {0, 0};
  hipEvent_t ctr_ev[2] = {nullptr, nullptr};                // upload of the buffer done
  hipEvent_t ctr_rd[2][kAux] = {};                          // last pre-pass reader of the buffer, per pre-pass stream
  bool ctr_rd_set[2] = {false, false};
  int ctr_parity = 0, ctr_cur = 0;
  uint16_t* depth_stage[kDepth] = {};              // host frames of the batch in flight, by pipeline slot
  uint32_t *zbuf[kAux] = {}, *lastzero[kAux] = {}, *zfix[kAux] = {};  // Reproject's z-buffer and replay state, one per pre-pass stream
  double *T12 = nullptr, *seg12 = nullptr, *madj12 = nullptr, *dsum = nullptr;
  int *grid_index = nullptr, *key_scratch = nullptr, *slot_scratch = nullptr;
  PlanRec* plan_rec[kDepth] = {};
  Plan* plan[kDepth] = {};
  bool reset_pending[kDepth] = {};                  // k_reset of the slot's last batch has not been launched yet
  size_t key_scratch_cap = 0;
  // profiling
  int prof_stride = 0;                              // 0 = off, n = time every n-th k_integrate launch
  long prof_tick = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  double ms_total = 0.0;
  long launches = 0, frames_done = 0;
  // round 6 (owner merge): units this GPU handed to their owner -- zeroed, still in the table, hidden from every key / count / extraction query until
  // the next frame is integrated or the unit is imported again -- and the grow-only device scratch of the band kernels
  std::vector<int> dropped;                         // sorted
  void* band_scratch = nullptr;
  size_t band_scratch_cap = 0;
};

hipStream_t er::tsdf_stream(er_tsdf_s* h) { return h->stream; }
int er::tsdf_device(er_tsdf_s* h) { return h->device; }

static int launch_reproject(er_tsdf_t h, const ReprojArgs& RA, int n, hipStream_t X);

namespace {

int check_flags(er_tsdf_t h) {
  int c[C_COUNT];
  ER_HIP_TRY(hipMemcpyAsync(c, h->counters, sizeof c, hipMemcpyDeviceToHost, h->stream));
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  if (c[C_POOL_OVERFLOW])
    return er::fail("TSDF unit pool exhausted: %d units requested, capacity %d (raise max_units)", c[C_NUNITS], h->max_units);
  if (c[C_TABLE_FULL]) return er::fail("TSDF unit hash table full (capacity %d)", h->ht_cap);
  return 0;
}

int drain_events(er_tsdf_t h) {
  for (auto& ev : h->events) {
    float ms = 0.f;
    ER_HIP_TRY(hipEventSynchronize(ev.second));
    ER_HIP_TRY(hipEventElapsedTime(&ms, ev.first, ev.second));
    h->ms_total += ms;
    h->launches += 1;
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  h->events.clear();
  return 0;
}

int ensure_key_scratch(er_tsdf_t h, size_t n) {
  if (n <= h->key_scratch_cap) return 0;
  if (h->key_scratch) (void)hipFree(h->key_scratch);
  if (h->slot_scratch) (void)hipFree(h->slot_scratch);
  h->key_scratch = h->slot_scratch = nullptr;
  size_t cap = std::max<size_t>(n, 1024);
  ER_HIP_TRY(hipMalloc(&h->key_scratch, cap * sizeof(int)));
  ER_HIP_TRY(hipMalloc(&h->slot_scratch, cap * sizeof(int)));
  h->key_scratch_cap = cap;
  return 0;
}

// keys (host) -> slots (device slot_scratch), optionally allocating missing units.
int resolve_slots(er_tsdf_t h, const int* keys_host, int n, bool allocate) {
  if (ensure_key_scratch(h, (size_t)n)) return 1;
  ER_HIP_TRY(hipMemcpyAsync(h->key_scratch, keys_host, (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_ensure_units, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, h->stream, h->key_scratch, n,
                     h->ht_key, h->ht_slot, h->ht_cap - 1, h->ht_shift, h->unit_key, h->max_units, h->counters,
                     h->slot_scratch, allocate ? 1 : 0);
  ER_HIP_TRY(hipGetLastError());
  return 0;
}

int sorted_units(er_tsdf_t h, std::vector<int>& keys, std::vector<int>& slots) {
  if (check_flags(h)) return 1;
  int n = 0;
  ER_HIP_TRY(hipMemcpy(&n, h->counters + C_NUNITS, sizeof(int), hipMemcpyDeviceToHost));
  n = std::min(n, h->max_units);
  std::vector<int> uk((size_t)n);
  if (n) ER_HIP_TRY(hipMemcpy(uk.data(), h->unit_key, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
  std::vector<std::pair<int, int>> ks;
  ks.reserve((size_t)n);
  for (int s = 0; s < n; s++)
    if (!std::binary_search(h->dropped.begin(), h->dropped.end(), uk[(size_t)s])) ks.push_back(std::make_pair(uk[(size_t)s], s));   // (handed to their owner)
  std::sort(ks.begin(), ks.end());
  n = (int)ks.size();
  keys.resize((size_t)n);
  slots.resize((size_t)n);
  for (int s = 0; s < n; s++) {
    keys[(size_t)s] = ks[(size_t)s].first;
    slots[(size_t)s] = ks[(size_t)s].second;
  }
  return 0;
}

int ensure_band_scratch(er_tsdf_t h, size_t bytes) {
  if (bytes <= h->band_scratch_cap) return 0;
  if (h->band_scratch) (void)hipFree(h->band_scratch);
  h->band_scratch = nullptr;
  h->band_scratch_cap = 0;
  const size_t cap = std::max<size_t>(bytes, (size_t)1 << 20);
  ER_HIP_TRY(hipMalloc(&h->band_scratch, cap));
  h->band_scratch_cap = cap;
  return 0;
}

void undrop(er_tsdf_t h, const int* keys, int n) {
  if (h->dropped.empty()) return;
  for (int i = 0; i < n; i++) {
    auto it = std::lower_bound(h->dropped.begin(), h->dropped.end(), keys[i]);
    if (it != h->dropped.end() && *it == keys[i]) h->dropped.erase(it);
  }
}

// Host staging layout of one batch's constants inside the pinned buffer of its parity.
struct Staging {
  er::FrameXform fx[ER_MAX_BATCH];
  float fxT[16][ER_MAX_BATCH];                     // the same constants component-major, DIRECTLY behind fx (k_integrate's culling reads them with lane = frame)
  double t12[ER_MAX_BATCH * 12];
  double seg[ER_MAX_BATCH * 16];
  double madj[ER_MAX_BATCH * 12];
  int gi[ER_MAX_BATCH];
};

// k_reset of a batch (clears the frame masks of its unit list, accounts the unit visits) is deferred: it runs on the pre-pass
// stream of the batch that reuses the slot, off the main stream, whose per-batch chain is then ONE kernel.  Whoever needs the
// accounts or leaves the pipeline (synchronise, profile read-out) flushes the pending ones on the main stream.
int flush_resets(er_tsdf_t h) {
  for (int q = 0; q < kDepth; q++)
    if (h->reset_pending[q]) {
      hipLaunchKernelGGL(k_reset, dim3(1), dim3(kBlock), 0, h->stream, h->batch[q], h->counters + kNbatchSlot[q], h->ht_mask[q], h->stats);
      ER_HIP_TRY(hipGetLastError());
      ER_HIP_TRY(hipEventRecord(h->int_done[q], h->stream));            // the slot's next user waits for this reset as well
      h->reset_pending[q] = false;
    }
  return 0;
}

int sync_all(er_tsdf_t h) {
  if (flush_resets(h)) return 1;
  if (h->copy_stream) ER_HIP_TRY(hipStreamSynchronize(h->copy_stream));
  for (int a = 0; a < kAux; a++) ER_HIP_TRY(hipStreamSynchronize(h->aux_stream[a]));
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

// One batch (<= ER_MAX_BATCH frames).  depth_dev: n * pixels uint16 on device (must be complete: the
// pre-passes run on the handle's auxiliary streams, which do not wait for the caller's stream).
// Pipeline over three in-order streams, batch state triple-buffered by slot p = batch mod 3:
//   aux stream b mod 2: wait int_done[p] -> [k_reset(p) of batch n-3] -> [H2D constants] -> k_reproject_* -> k_prepare(p) -> k_plan(p) -> event pre_done[p]
//   main stream       : wait pre_done[p] -> k_integrate(p) -> event int_done[p]            (ONE kernel per batch: it is the critical path)
// so the pre-passes of batches n+1 and n+2 (latency / float64 bound, each a serial chain of launches) overlap each other and
// k_integrate of batch n (float32 VALU bound): k_integrate is shorter than one pre-pass chain, so two chains run side by side.
// Batches still reach the volume strictly in order (main stream).
int run_batch(er_tsdf_t h, int n, const uint16_t* depth_dev, const double* T, const er_warp* warp, int frame0) {
  const int p = (int)(h->batch_no % kDepth), a = (int)(h->batch_no % kAux);
  h->batch_no++;
  // Slot p was last used by batch n-3.  The HOST only needs its pinned constants block back (the H2D copy of batch n-3, an
  // early event); the DEVICE buffers of the slot (scaled depth, masks, frame constants) are protected on the device: the
  // pre-pass stream waits for k_integrate of batch n-3 before it touches them.  The host therefore never blocks on a voxel
  // pass and runs up to three batches ahead (host-frame copies and pre-passes queue up behind the events).
  if (h->used[p]) ER_HIP_TRY(hipEventSynchronize(h->consts_done[p]));
  Staging* st = static_cast<Staging*>(h->pinned[p]);
  for (int f = 0; f < n; f++) {
    const double* Tf = T + (size_t)f * 16;
    double Tinv[16];
    if (!er::mat4_inverse(Tf, Tinv)) return er::fail("frame %d: singular pose matrix", frame0 + f);
    for (int q = 0; q < 12; q++) {
      st->fx[f].mi[q] = (float)Tinv[q];                          // trans_inv.cast<float>(), TSDFVolume.cpp:59
      st->t12[f * 12 + q] = Tf[q];
    }
    st->fx[f].tx = (float)Tf[3];                                 // transformation.cast<float>()(r,3)
    st->fx[f].ty = (float)Tf[7];
    st->fx[f].tz = (float)Tf[11];
    st->fx[f].pad = 0.f;
    for (int q = 0; q < 16; q++) st->fxT[q][f] = reinterpret_cast<const float*>(&st->fx[f])[q];
  }
  static_assert(offsetof(Staging, fxT) == sizeof(er::FrameXform) * ER_MAX_BATCH && sizeof(er::FrameXform) == 64, "fxT lies directly behind fx");
  hipStream_t X = h->aux_stream[a], S = h->stream;
  int* nbatch = h->counters + kNbatchSlot[p];

#ifndef ER_INT_BLOCKS_PER_CU
#define ER_INT_BLOCKS_PER_CU 3
#endif
  constexpr int kIntBlocksPerCu = ER_INT_BLOCKS_PER_CU;       // persistent workgroups fed by the queue.  Fewer than fit: the pre-pass kernels need register
                                           // space next to them (2 -> 172.0 k, 3 -> 174.3 k, 4 -> 170.0 k, 5 -> 169.6 k frames/s, profiles/r05n_*)
  const int wide_grid = h->n_cu * kIntBlocksPerCu;
  uint32_t* zsrc = nullptr;
  char* dst = static_cast<char*>(h->dstage[p]);
  const double* dev_t12 = reinterpret_cast<const double*>(dst + offsetof(Staging, t12));
  const double* dev_seg = reinterpret_cast<const double*>(dst + offsetof(Staging, seg));
  const double* dev_madj = reinterpret_cast<const double*>(dst + offsetof(Staging, madj));
  const int* dev_gi = reinterpret_cast<const int*>(dst + offsetof(Staging, gi));
  if (warp) {
    for (int f = 0; f < n; f++) {
      for (int q = 0; q < 12; q++) {
        st->seg[f * 16 + q] = warp->seg[(size_t)(frame0 + f) * 16 + q];
        st->madj[f * 12 + q] = warp->madj[(size_t)(frame0 + f) * 16 + q];
      }
      er::cube_coord_deltas(&st->seg[f * 16], h->cam, h->cols, h->rows, &st->seg[f * 16 + 12]);
      st->seg[f * 16 + 15] = 0.0;
      const int g = warp->grid_index[frame0 + f];
      if (g < 0 || g >= warp->num_grids) return er::fail("frame %d: control grid index %d out of [0,%d)", frame0 + f, g, warp->num_grids);
      st->gi[f] = g;
    }
  }
  // all per-batch constants travel in ONE copy (every launch or copy on this stream costs ~5 us of the pre-pass chain)
  if (h->used[p]) ER_HIP_TRY(hipStreamWaitEvent(X, h->int_done[p], 0));     // k_integrate of batch n-3 still reads dstage[p] / scaled[p] / masks[p]
  if (h->reset_pending[p]) {
    hipLaunchKernelGGL(k_reset, dim3(1), dim3(kBlock), 0, X, h->batch[p], nbatch, h->ht_mask[p], h->stats);
    ER_HIP_TRY(hipGetLastError());
    h->reset_pending[p] = false;
  }
  ER_HIP_TRY(hipMemcpyAsync(h->dstage[p], st, sizeof(Staging), hipMemcpyHostToDevice, X));
  ER_HIP_TRY(hipEventRecord(h->consts_done[p], X));
  if (warp) {
    // zbuf is all-empty here: filled at create, re-armed by its consumer (k_prepare / k_zbuf_to_depth)
    const int verts = (warp->resolution + 1) * (warp->resolution + 1) * (warp->resolution + 1);
    const float grid_ul = warp->length / (float)warp->resolution;       // ControlGrid.cpp:19
    const ReprojArgs RA{depth_dev, n, h->cols, h->rows, h->cam, h->cami, dev_seg, dev_madj, dev_gi, h->ctr, warp->resolution, grid_ul,
                        verts * 3, h->zbuf[a], h->lastzero[a], h->zfix[a], h->counters + kZeroFlagSlot[a]};
    if (launch_reproject(h, RA, n, X)) return 1;
    zsrc = h->zbuf[a];
  }

  hipLaunchKernelGGL(k_prepare, dim3((h->cols + kTile - 1) / kTile, (h->rows + kTile - 1) / kTile, n), dim3(kPrepThreads), 0, X,
                     depth_dev, zsrc, n, h->cols, h->rows, h->cam, h->cami, h->lambda, dev_t12, h->scaled[p], h->ht_key, h->ht_slot,
                     h->ht_mask[p], h->ht_cap - 1, h->ht_shift, h->batch[p], nbatch, h->counters,
                     h->tile_max[p], h->tile_lo[p], h->tile_lo_fine[p], make_int2(h->shard_rank, h->shard_world), h->lastzero[a], h->zfix[a],
                     h->counters + kZeroFlagSlot[a]);
  hipLaunchKernelGGL(k_plan, dim3(1), dim3(256), 0, X, h->batch[p], nbatch, h->ht_mask[p], h->ht_key, h->ht_slot, h->unit_key, h->max_units,
                     h->counters, h->plan_rec[p], h->plan[p], warp ? h->counters + kZeroFlagSlot[a] : (int*)nullptr);
  ER_HIP_TRY(hipGetLastError());
  ER_HIP_TRY(hipEventRecord(h->pre_done[p], X));

  ER_HIP_TRY(hipStreamWaitEvent(S, h->pre_done[p], 0));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool timed = h->prof_stride > 0 && (h->prof_tick++ % h->prof_stride) == 0;
  if (timed) {
    ER_HIP_TRY(hipEventCreate(&e0));
    ER_HIP_TRY(hipEventCreate(&e1));
    ER_HIP_TRY(hipEventRecord(e0, S));
  }
  const bool sure = h->cam.integration_trunc < 64.0f;                   // voxel_classify's bound on the scaled depth (false for NaN)
  hipLaunchKernelGGL(sure ? k_integrate<true> : k_integrate<false>, dim3(wide_grid), dim3(kBlock), 0, S, h->pool, h->plan_rec[p], h->plan[p],
                     h->frames[p], h->scaled[p], h->tile_max[p], h->tile_lo[p], h->tile_lo_fine[p], (h->cols + kTile - 1) / kTile, (h->rows + kTile - 1) / kTile,
                     h->cam, h->cols, h->rows);
  if (timed) {
    ER_HIP_TRY(hipEventRecord(e1, S));
    h->events.emplace_back(e0, e1);
  }
  ER_HIP_TRY(hipGetLastError());
  ER_HIP_TRY(hipEventRecord(h->int_done[p], S));
  h->reset_pending[p] = true;                               // (launched by the slot's next user, or by flush_resets)
  h->used[p] = true;
  h->frames_done += n;
  return 0;
}

}  // namespace

static hipError_t aux_create(hipStream_t* s) {
  // (stream priorities for the pre-pass streams, lowest or highest against the voxel stream's default: no effect on the job or on
  //  k_integrate's time in the pipeline, profiles/r03r_ab_stream_priority.txt)
  return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

extern "C" {

int er_tsdf_create(int cols, int rows, const float cam6[6], int max_units, int device, er_tsdf_t* out) {
  if (!out) return er::fail("er_tsdf_create: out is NULL");
  *out = nullptr;
  if (cols <= 0 || rows <= 0 || max_units <= 0) return er::fail("er_tsdf_create: bad dimensions");
  if ((long)cols * rows >= (1L << 30)) return er::fail("er_tsdf_create: image of %d x %d pixels is too large", cols, rows);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return er::fail("er_tsdf_create: no HIP device available (liber_hip has no CPU fallback)");
  if (device < 0 || device >= ndev) return er::fail("er_tsdf_create: device %d out of range [0,%d)", device, ndev);
  ER_HIP_TRY(hipSetDevice(device));
  er_tsdf_t h = new er_tsdf_s();
  h->device = device;
  h->cols = cols;
  h->rows = rows;
  h->pixels = cols * rows;
  h->max_units = max_units;
  if (cam6) {
    h->cam = er::Camera{cam6[0], cam6[1], cam6[2], cam6[3], cam6[4], cam6[5]};
  } else {
    h->cam = er::Camera{525.0f, 525.0f, 319.5f, 239.5f, 2.5f, 2.5f};   // TSDFVolumeUnit.h:69
  }
  h->cami.inv_fx = 1.0 / (double)h->cam.fx;
  h->cami.inv_fy = 1.0 / (double)h->cam.fy;
  h->cami.pp_small = (std::fabs((double)h->cam.cx) < 1e6 && std::fabs((double)h->cam.cy) < 1e6) ? 1 : 0;
  h->cami.pad = 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) h->n_cu = prop.multiProcessorCount;
  int cap = 1024, lg = 10;
  while (cap < 4 * max_units) { cap <<= 1; lg++; }
  h->ht_cap = cap;
  h->ht_shift = 32 - lg;
  const size_t px = (size_t)h->pixels, B = ER_MAX_BATCH;
#define ER_ALLOC(ptr, bytes)                                                                         \
  do {                                                                                               \
    hipError_t e_ = hipMalloc((void**)&(ptr), (bytes));                                              \
    if (e_ != hipSuccess) {                                                                          \
      er::fail("er_tsdf_create: hipMalloc(%zu bytes) for " #ptr " failed: %s", (size_t)(bytes), hipGetErrorString(e_)); \
      er_tsdf_destroy(h);                                                                            \
      return 1;                                                                                      \
    }                                                                                                \
  } while (0)
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess ||
      aux_create(&h->aux_stream[0]) != hipSuccess || (kAux > 1 && aux_create(&h->aux_stream[kAux - 1]) != hipSuccess) ||
      false) {
    delete h;
    return er::fail("er_tsdf_create: hipStreamCreate failed");
  }
  h->stream = h->own_stream;
  for (int q = 0; q < kDepth; q++) {
    if (hipEventCreateWithFlags(&h->pre_done[q], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->int_done[q], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->copy_done[q], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->consts_done[q], hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc(&h->pinned[q], sizeof(Staging), hipHostMallocDefault) != hipSuccess) {
      er_tsdf_destroy(h);
      return er::fail("er_tsdf_create: event / pinned staging allocation failed");
    }
  }
  ER_ALLOC(h->pool, (size_t)max_units * er::kUnitVox * sizeof(float2));
  ER_ALLOC(h->ht_key, (size_t)cap * sizeof(int));
  ER_ALLOC(h->ht_slot, (size_t)cap * sizeof(int));
  for (int q = 0; q < kDepth; q++) ER_ALLOC(h->ht_mask[q], (size_t)cap * sizeof(unsigned long long));
  ER_ALLOC(h->unit_key, (size_t)max_units * sizeof(int));
  ER_ALLOC(h->counters, C_COUNT * sizeof(int));
  ER_ALLOC(h->stats, 4 * sizeof(unsigned long long));
  for (int q = 0; q < kDepth; q++) ER_ALLOC(h->batch[q], (size_t)cap * sizeof(int));
  ER_ALLOC(h->lambda, px * sizeof(float));
  for (int q = 0; q < kDepth; q++) ER_ALLOC(h->scaled[q], B * (px + kScaledPad) * sizeof(float));
  for (int q = 0; q < kDepth; q++) ER_ALLOC(h->depth_stage[q], B * px * sizeof(uint16_t));
  for (int q = 0; q < kAux; q++) ER_ALLOC(h->zbuf[q], B * px * sizeof(uint32_t));
  for (int q = 0; q < kAux; q++) ER_ALLOC(h->lastzero[q], B * px * sizeof(uint32_t));
  for (int q = 0; q < kAux; q++) ER_ALLOC(h->zfix[q], B * px * sizeof(uint32_t));
  for (int q = 0; q < kDepth; q++) ER_ALLOC(h->dstage[q], sizeof(Staging));   // device twin of the pinned staging block: ONE copy per batch
  for (int q = 0; q < kDepth; q++) h->frames[q] = reinterpret_cast<er::FrameXform*>(reinterpret_cast<char*>(h->dstage[q]) + offsetof(Staging, fx));
  ER_ALLOC(h->T12, B * 12 * sizeof(double));
  ER_ALLOC(h->seg12, B * 16 * sizeof(double));
  ER_ALLOC(h->madj12, B * 12 * sizeof(double));
  ER_ALLOC(h->grid_index, B * sizeof(int));
  ER_ALLOC(h->dsum, sizeof(double));
  for (int q = 0; q < kDepth; q++) ER_ALLOC(h->tile_max[q], B * (size_t)((cols + kTile - 1) / kTile) * ((rows + kTile - 1) / kTile) * sizeof(float));
  for (int q = 0; q < kDepth; q++) ER_ALLOC(h->tile_lo[q], B * (size_t)((cols + kTile - 1) / kTile) * ((rows + kTile - 1) / kTile) * sizeof(float));
  for (int q = 0; q < kDepth; q++)
    ER_ALLOC(h->tile_lo_fine[q], B * (size_t)((cols + (1 << kLoShift) - 1) >> kLoShift) * ((rows + (1 << kLoShift) - 1) >> kLoShift) * sizeof(float));
  for (int q = 0; q < kDepth; q++) ER_ALLOC(h->plan_rec[q], (size_t)cap * sizeof(PlanRec));
  for (int q = 0; q < kDepth; q++) ER_ALLOC(h->plan[q], sizeof(Plan));
#undef ER_ALLOC
  hipStream_t s = h->stream;
  bool ok = hipMemsetAsync(h->pool, 0, (size_t)max_units * er::kUnitVox * sizeof(float2), s) == hipSuccess &&
            hipMemsetAsync(h->ht_key, 0xFF, (size_t)cap * sizeof(int), s) == hipSuccess &&
            hipMemsetAsync(h->ht_slot, 0xFF, (size_t)cap * sizeof(int), s) == hipSuccess &&
            hipMemsetAsync(h->counters, 0, C_COUNT * sizeof(int), s) == hipSuccess &&
            hipMemsetAsync(h->stats, 0, 4 * sizeof(unsigned long long), s) == hipSuccess;
  for (int q = 0; q < kDepth; q++)
    ok = ok && hipMemsetAsync(h->ht_mask[q], 0, (size_t)cap * sizeof(unsigned long long), s) == hipSuccess &&
         hipMemsetAsync(h->scaled[q], 0, B * (px + kScaledPad) * sizeof(float), s) == hipSuccess;   // (the pads stay zero)
  for (int q = 0; q < kAux; q++)
    ok = ok && hipMemsetAsync(h->lastzero[q], 0, B * px * sizeof(uint32_t), s) == hipSuccess &&
         hipMemsetAsync(h->zbuf[q], 0xFF, B * px * sizeof(uint32_t), s) == hipSuccess &&
         hipMemsetAsync(h->zfix[q], 0xFF, B * px * sizeof(uint32_t), s) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(k_lambda, dim3((h->pixels + kBlock - 1) / kBlock), dim3(kBlock), 0, s, h->lambda, cols, rows, h->cam);
    ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  }
  if (!ok) {
    er_tsdf_destroy(h);
    return er::fail("er_tsdf_create: device initialisation failed: %s", hipGetErrorString(hipGetLastError()));
  }
  *out = h;
  return 0;
}

int er_tsdf_destroy(er_tsdf_t h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (auto& ev : h->events) {
    (void)hipEventDestroy(ev.first);
    (void)hipEventDestroy(ev.second);
  }
  for (int a = 0; a < kAux; a++)
    if (h->aux_stream[a]) (void)hipStreamSynchronize(h->aux_stream[a]);
  std::vector<void*> ptrs = {h->pool, h->ht_key, h->ht_slot, h->unit_key, h->counters, h->stats, h->lambda, h->T12, h->seg12, h->madj12,
                             h->grid_index, h->dsum, h->ctr_dev[0], h->ctr_dev[1], h->key_scratch,
                             h->slot_scratch, h->band_scratch};
  for (int q = 0; q < kDepth; q++)
    for (void* x : {(void*)h->ht_mask[q], (void*)h->batch[q], (void*)h->scaled[q], (void*)h->depth_stage[q], h->dstage[q], (void*)h->tile_max[q], (void*)h->tile_lo[q], (void*)h->tile_lo_fine[q],
                    (void*)h->plan_rec[q], (void*)h->plan[q]})
      ptrs.push_back(x);
  for (int q = 0; q < kAux; q++) {
    ptrs.push_back(h->zbuf[q]);
    ptrs.push_back(h->lastzero[q]);
    ptrs.push_back(h->zfix[q]);
  }
  for (int q = 0; q < kDepth; q++) {
    if (h->pre_done[q]) (void)hipEventDestroy(h->pre_done[q]);
    if (h->int_done[q]) (void)hipEventDestroy(h->int_done[q]);
    if (h->copy_done[q]) (void)hipEventDestroy(h->copy_done[q]);
    if (h->consts_done[q]) (void)hipEventDestroy(h->consts_done[q]);
    if (h->pinned[q]) (void)hipHostFree(h->pinned[q]);
  }
  for (int a = 0; a < kAux; a++)
    if (h->aux_stream[a]) (void)hipStreamDestroy(h->aux_stream[a]);
  if (h->copy_stream) {
    (void)hipStreamSynchronize(h->copy_stream);
    (void)hipStreamDestroy(h->copy_stream);
  }
  for (int q = 0; q < 2; q++) {
    if (h->ctr_ev[q]) (void)hipEventDestroy(h->ctr_ev[q]);
    for (int a = 0; a < kAux; a++)
      if (h->ctr_rd[q][a]) (void)hipEventDestroy(h->ctr_rd[q][a]);
    if (h->ctr_pinned[q]) (void)hipHostFree(h->ctr_pinned[q]);
  }
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return 0;
}

int er_tsdf_set_stream(er_tsdf_t h, void* hip_stream) {
  if (!h) return er::fail("er_tsdf_set_stream: NULL handle");
  ER_HIP_TRY(hipSetDevice(h->device));
  if (sync_all(h)) return 1;
  h->stream = hip_stream ? (hipStream_t)hip_stream : h->own_stream;
  return 0;
}

int er_tsdf_synchronize(er_tsdf_t h) {
  if (!h) return er::fail("er_tsdf_synchronize: NULL handle");
  ER_HIP_TRY(hipSetDevice(h->device));
  return sync_all(h);
}

int er_tsdf_scale_depth(er_tsdf_t h, const uint16_t* depth_host, float* scaled_host) {
  if (!h || !depth_host || !scaled_host) return er::fail("er_tsdf_scale_depth: NULL argument");
  ER_HIP_TRY(hipSetDevice(h->device));
  if (sync_all(h)) return 1;
  const size_t px = (size_t)h->pixels;
  ER_HIP_TRY(hipMemcpyAsync(h->depth_stage[0], depth_host, px * sizeof(uint16_t), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_scale_depth, dim3((h->pixels + kBlock - 1) / kBlock), dim3(kBlock), 0, h->stream, h->depth_stage[0],
                     h->lambda, h->scaled[0], h->pixels, h->cam.integration_trunc);
  ER_HIP_TRY(hipGetLastError());
  ER_HIP_TRY(hipMemcpyAsync(scaled_host, h->scaled[0], px * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

// num_grids lattices of (res+1)^3 x 3 floats -> device.
// The copy runs on the pre-pass stream of the call's first batch into the device buffer of this call's parity (so it never
// waits for the previous call's pre-passes, which read the other buffer); the other pre-pass stream (and `also`, if given)
// waits for it.  Sets h->ctr to the buffer the kernels of this call read.
static int upload_ctr(er_tsdf_t h, const float* ctr, int res, int num_grids, hipStream_t also) {
  const size_t verts = (size_t)(res + 1) * (res + 1) * (res + 1);
  const size_t floats = verts * 3 * (size_t)num_grids;
  const int q = h->ctr_parity;
  h->ctr_parity ^= 1;
  if (floats > h->ctr_dev_cap[q]) {
    if (sync_all(h)) return 1;                       // nobody may still be reading the old buffer
    if (h->ctr_dev[q]) (void)hipFree(h->ctr_dev[q]);
    h->ctr_dev[q] = nullptr;
    h->ctr_dev_cap[q] = 0;
    ER_HIP_TRY(hipMalloc((void**)&h->ctr_dev[q], floats * sizeof(float)));
    h->ctr_dev_cap[q] = floats;
  }
  // The lattices travel through a page-locked block of the handle: the copy is then truly asynchronous (a copy from the
  // caller's pageable memory would make the host wait for everything queued on this stream at every call) and the caller's
  // memory is free again when the call returns.
  if (!h->ctr_ev[q]) {
    ER_HIP_TRY(hipEventCreateWithFlags(&h->ctr_ev[q], hipEventDisableTiming));
    for (int a = 0; a < kAux; a++) ER_HIP_TRY(hipEventCreateWithFlags(&h->ctr_rd[q][a], hipEventDisableTiming));
  } else {
    ER_HIP_TRY(hipEventSynchronize(h->ctr_ev[q]));          // the upload of two calls ago (pinned block free again)
  }
  if (floats > h->ctr_pinned_cap[q]) {
    if (h->ctr_pinned[q]) (void)hipHostFree(h->ctr_pinned[q]);
    h->ctr_pinned[q] = nullptr;
    h->ctr_pinned_cap[q] = 0;
    ER_HIP_TRY(hipHostMalloc((void**)&h->ctr_pinned[q], floats * sizeof(float), hipHostMallocDefault));
    h->ctr_pinned_cap[q] = floats;
  }
  memcpy(h->ctr_pinned[q], ctr, floats * sizeof(float));
  const int a0 = (int)(h->batch_no % kAux);
  hipStream_t C = h->aux_stream[a0];
  if (h->ctr_rd_set[q])                                     // the pre-passes of two calls ago read this device buffer
    for (int a = 0; a < kAux; a++)
      if (a != a0) ER_HIP_TRY(hipStreamWaitEvent(C, h->ctr_rd[q][a], 0));
  ER_HIP_TRY(hipMemcpyAsync(h->ctr_dev[q], h->ctr_pinned[q], floats * sizeof(float), hipMemcpyHostToDevice, C));
  ER_HIP_TRY(hipEventRecord(h->ctr_ev[q], C));
  for (int a = 0; a < kAux; a++)
    if (a != a0) ER_HIP_TRY(hipStreamWaitEvent(h->aux_stream[a], h->ctr_ev[q], 0));
  if (also) ER_HIP_TRY(hipStreamWaitEvent(also, h->ctr_ev[q], 0));
  h->ctr = h->ctr_dev[q];
  h->ctr_cur = q;
  return 0;
}

// Reproject of n frames into zbuf: the all-exact kernel, or (-DER_REPROJECT_TIERED) tier 1 + the exact tail; then the replay launch.
static int launch_reproject(er_tsdf_t h, const ReprojArgs& RA, int n, hipStream_t X) {
  // (staging the lattice in LDS for this kernel was measured: slower, profiles/r02d_ab_lds_lattice.txt; a float32 tier with a
  //  per-pixel proof in front of the exact chain, four designs: slower, profiles/r02b / r02c / r02z_ab_tiered_reproject_*.txt)
  hipLaunchKernelGGL(k_reproject_scatter, dim3((h->cols + 63) / 64, (h->rows + 3) / 4, n), dim3(kBlock), 0, X, RA);
  hipLaunchKernelGGL(k_reproject_fix, dim3(kFixBlocks), dim3(kFixThreads), 0, X, RA);   // single waves: they have to find room next to three busy kernels
  ER_HIP_TRY(hipGetLastError());
  return 0;
}

int er_tsdf_reproject(er_tsdf_t h, uint16_t* depth_inout_host, const float* ctr_host, int resolution, float length,
                      const double seg[16], const double madj[16]) {
  if (!h || !depth_inout_host || !ctr_host || !seg || !madj) return er::fail("er_tsdf_reproject: NULL argument");
  if (resolution <= 0) return er::fail("er_tsdf_reproject: bad resolution");
  ER_HIP_TRY(hipSetDevice(h->device));
  const size_t px = (size_t)h->pixels;
  const int verts = (resolution + 1) * (resolution + 1) * (resolution + 1);
  if (sync_all(h)) return 1;                         // single-frame hook: runs alone on the main stream
  if (upload_ctr(h, ctr_host, resolution, 1, h->stream)) return 1;      // (on a pre-pass stream; the main stream waits for it)
  const int gi = 0;
  ER_HIP_TRY(hipMemcpyAsync(h->depth_stage[0], depth_inout_host, px * sizeof(uint16_t), hipMemcpyHostToDevice, h->stream));
  double seg16[16] = {0};
  memcpy(seg16, seg, 12 * sizeof(double));
  er::cube_coord_deltas(seg16, h->cam, h->cols, h->rows, seg16 + 12);
  ER_HIP_TRY(hipMemcpyAsync(h->seg12, seg16, 16 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  ER_HIP_TRY(hipMemcpyAsync(h->madj12, madj, 12 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  ER_HIP_TRY(hipMemcpyAsync(h->grid_index, &gi, sizeof(int), hipMemcpyHostToDevice, h->stream));
  const float grid_ul = length / (float)resolution;
  const long total = (long)px;
  const ReprojArgs RA{h->depth_stage[0], 1, h->cols, h->rows, h->cam, h->cami, h->seg12, h->madj12, h->grid_index, h->ctr, resolution, grid_ul,
                      verts * 3, h->zbuf[0], h->lastzero[0], h->zfix[0], h->counters + kZeroFlagSlot[0]};
  if (launch_reproject(h, RA, 1, h->stream)) return 1;
  hipLaunchKernelGGL(k_zbuf_to_depth, dim3((int)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, h->zbuf[0],
                     h->depth_stage[0], total, h->lastzero[0], h->zfix[0], h->counters + kZeroFlagSlot[0]);
  ER_HIP_TRY(hipGetLastError());
  ER_HIP_TRY(hipMemsetAsync(h->counters + kZeroFlagSlot[0], 0, 2 * sizeof(int), h->stream));
  ER_HIP_TRY(hipMemcpyAsync(depth_inout_host, h->depth_stage[0], px * sizeof(uint16_t), hipMemcpyDeviceToHost, h->stream));
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

int er_tsdf_integrate_frames(er_tsdf_t h, int n, const uint16_t* depth, int depth_on_device, const double* T,
                             const er_warp* warp) {
  if (!h || !depth || !T) return er::fail("er_tsdf_integrate_frames: NULL argument");
  if (n < 0) return er::fail("er_tsdf_integrate_frames: negative frame count");
  ER_HIP_TRY(hipSetDevice(h->device));
  if (n > 0) h->dropped.clear();                              // (handed-over units are zeroed: from here on they are this GPU's new contribution)
  if (warp) {
    if (!warp->ctr || !warp->grid_index || !warp->seg || !warp->madj || warp->num_grids <= 0 || warp->resolution <= 0)
      return er::fail("er_tsdf_integrate_frames: incomplete er_warp");
    if (upload_ctr(h, warp->ctr, warp->resolution, warp->num_grids, nullptr)) return 1;
  }
  const size_t px = (size_t)h->pixels;
  // n frames are fused in ceil(n / ER_MAX_BATCH) launches of (nearly) EQUAL size: 150 frames run as 3 x 50, not 64 + 64 + 22
  // (a voxel is loaded once per launch, so the short tail launch would pay the full volume traffic for a third of the frames)
  const int launches = (n + ER_MAX_BATCH - 1) / ER_MAX_BATCH;
  const int per = launches > 0 ? (n + launches - 1) / launches : 0;
  for (int start = 0; start < n; start += per) {
    const int nb = std::min(per, n - start);
    const uint16_t* ddev;
    if (depth_on_device) {
      ddev = depth + (size_t)start * px;
    } else {
      // Host frames travel on their own stream into the staging buffer of this batch's slot: the copy of batch n+1 overlaps
      // the pre-passes of the batches before it (aux streams) and the voxel pass (main stream) when the caller's memory is
      // page-locked (er_host_alloc); pageable memory makes hipMemcpyAsync block the host, which is still correct.
      const int p = (int)(h->batch_no % kDepth);
      if (!h->copy_stream) ER_HIP_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
      if (h->used[p]) ER_HIP_TRY(hipStreamWaitEvent(h->copy_stream, h->pre_done[p], 0));   // the pre-pass that last read depth_stage[p] (batch n-3)
      ER_HIP_TRY(hipMemcpyAsync(h->depth_stage[p], depth + (size_t)start * px, (size_t)nb * px * sizeof(uint16_t),
                                hipMemcpyHostToDevice, h->copy_stream));
      ER_HIP_TRY(hipEventRecord(h->copy_done[p], h->copy_stream));
      ER_HIP_TRY(hipStreamWaitEvent(h->aux_stream[h->batch_no % kAux], h->copy_done[p], 0));
      ddev = h->depth_stage[p];
    }
    if (run_batch(h, nb, ddev, T + (size_t)start * 16, warp, start)) return 1;
  }
  if (warp) {                                               // the last readers of this call's lattice buffer, per pre-pass stream
    for (int a = 0; a < kAux; a++) ER_HIP_TRY(hipEventRecord(h->ctr_rd[h->ctr_cur][a], h->aux_stream[a]));
    h->ctr_rd_set[h->ctr_cur] = true;
  }
  // The caller may reuse or free its HOST frames as soon as the call returns: wait for the copies (not for the kernels).
  if (!depth_on_device && n > 0) ER_HIP_TRY(hipStreamSynchronize(h->copy_stream));
  return 0;
}

int er_tsdf_integrate(er_tsdf_t h, const uint16_t* depth_host, const double T[16]) {
  return er_tsdf_integrate_frames(h, 1, depth_host, 0, T, nullptr);
}

int er_tsdf_wait_event(er_tsdf_t h, void* hip_event) {
  if (!h || !hip_event) return er::fail("er_tsdf_wait_event: NULL argument");
  ER_HIP_TRY(hipSetDevice(h->device));
  for (int a = 0; a < kAux; a++) ER_HIP_TRY(hipStreamWaitEvent(h->aux_stream[a], (hipEvent_t)hip_event, 0));
  if (h->copy_stream) ER_HIP_TRY(hipStreamWaitEvent(h->copy_stream, (hipEvent_t)hip_event, 0));
  return 0;
}

int er_tsdf_reset(er_tsdf_t h) {
  if (!h) return er::fail("er_tsdf_reset: NULL handle");
  ER_HIP_TRY(hipSetDevice(h->device));
  if (sync_all(h)) return 1;
  int n = 0;
  ER_HIP_TRY(hipMemcpy(&n, h->counters + C_NUNITS, sizeof(int), hipMemcpyDeviceToHost));
  n = std::min(std::max(n, 0), h->max_units);
  hipStream_t s = h->stream;
  if (n > 0) ER_HIP_TRY(hipMemsetAsync(h->pool, 0, (size_t)n * er::kUnitVox * sizeof(float2), s));   // only the units ever handed out
  ER_HIP_TRY(hipMemsetAsync(h->ht_key, 0xFF, (size_t)h->ht_cap * sizeof(int), s));
  ER_HIP_TRY(hipMemsetAsync(h->ht_slot, 0xFF, (size_t)h->ht_cap * sizeof(int), s));
  for (int q = 0; q < kDepth; q++) ER_HIP_TRY(hipMemsetAsync(h->ht_mask[q], 0, (size_t)h->ht_cap * sizeof(unsigned long long), s));
  ER_HIP_TRY(hipMemsetAsync(h->counters, 0, C_COUNT * sizeof(int), s));
  ER_HIP_TRY(hipStreamSynchronize(s));
  for (int q = 0; q < kDepth; q++) h->used[q] = h->reset_pending[q] = false;
  h->batch_no = 0;
  h->dropped.clear();
  return 0;
}

int er_tsdf_set_unit_shard(er_tsdf_t h, int rank, int world) {
  if (!h) return er::fail("er_tsdf_set_unit_shard: NULL handle");
  if (world < 1 || rank < 0 || rank >= world) return er::fail("er_tsdf_set_unit_shard: rank %d not in [0,%d)", rank, world);
  ER_HIP_TRY(hipSetDevice(h->device));
  int n = 0;
  ER_HIP_TRY(hipMemcpy(&n, h->counters + C_NUNITS, sizeof(int), hipMemcpyDeviceToHost));
  if (n != 0) return er::fail("er_tsdf_set_unit_shard: the volume already holds %d units (set the shard before the first frame)", n);
  h->shard_rank = rank;
  h->shard_world = world;
  return 0;
}

int er_unit_owner(int key, int world) { return er::unit_owner(key, world); }

int er_tsdf_status(er_tsdf_t h, int* flags, long* out_of_range_pixels) {
  if (!h) return er::fail("er_tsdf_status: NULL handle");
  ER_HIP_TRY(hipSetDevice(h->device));
  int c[C_COUNT];
  ER_HIP_TRY(hipMemcpy(c, h->counters, sizeof c, hipMemcpyDeviceToHost));   // a poll: does not wait for the handle's (non-blocking) streams
  if (flags) *flags = (c[C_POOL_OVERFLOW] ? ER_STATUS_POOL_EXHAUSTED : 0) | (c[C_TABLE_FULL] ? ER_STATUS_TABLE_FULL : 0);
  if (out_of_range_pixels) *out_of_range_pixels = c[C_OUT_OF_RANGE];
  return 0;
}

int er_tsdf_unit_count(er_tsdf_t h, int* count) {
  if (!h || !count) return er::fail("er_tsdf_unit_count: NULL argument");
  ER_HIP_TRY(hipSetDevice(h->device));
  if (check_flags(h)) return 1;
  ER_HIP_TRY(hipMemcpy(count, h->counters + C_NUNITS, sizeof(int), hipMemcpyDeviceToHost));
  *count -= (int)h->dropped.size();                             // units handed to their owner by a distributed merge
  return 0;
}

int er_tsdf_unit_keys(er_tsdf_t h, int* keys_host) {
  if (!h || !keys_host) return er::fail("er_tsdf_unit_keys: NULL argument");
  ER_HIP_TRY(hipSetDevice(h->device));
  std::vector<int> keys, slots;
  if (sorted_units(h, keys, slots)) return 1;
  std::copy(keys.begin(), keys.end(), keys_host);
  return 0;
}

int er_tsdf_read_unit(er_tsdf_t h, int key, float* sdf_host, float* weight_host) {
  if (!h) return er::fail("er_tsdf_read_unit: NULL handle");
  ER_HIP_TRY(hipSetDevice(h->device));
  if (check_flags(h)) return 1;
  if (std::binary_search(h->dropped.begin(), h->dropped.end(), key)) return er::fail("er_tsdf_read_unit: unit %d was handed to its owner by the last merge", key);
  if (resolve_slots(h, &key, 1, false)) return 1;
  int slot = -1;
  ER_HIP_TRY(hipMemcpyAsync(&slot, h->slot_scratch, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  if (slot < 0) return er::fail("er_tsdf_read_unit: no unit with key %d", key);
  std::vector<float2> tmp((size_t)er::kUnitVox);
  ER_HIP_TRY(hipMemcpyAsync(tmp.data(), h->pool + (size_t)slot * er::kUnitVox, tmp.size() * sizeof(float2),
                            hipMemcpyDeviceToHost, h->stream));
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  for (int l = 0; l < er::kUnitVox; l++) {
    if (sdf_host) sdf_host[l] = tmp[(size_t)l].x;
    if (weight_host) weight_host[l] = tmp[(size_t)l].y;
  }
  return 0;
}

int er_tsdf_sum_weight(er_tsdf_t h, double* sum) {
  if (!h || !sum) return er::fail("er_tsdf_sum_weight: NULL argument");
  ER_HIP_TRY(hipSetDevice(h->device));
  if (check_flags(h)) return 1;
  int n = 0;
  ER_HIP_TRY(hipMemcpy(&n, h->counters + C_NUNITS, sizeof(int), hipMemcpyDeviceToHost));
  ER_HIP_TRY(hipMemsetAsync(h->dsum, 0, sizeof(double), h->stream));
  if (n > 0) {
    hipLaunchKernelGGL(k_sum_weight, dim3(h->n_cu * 8), dim3(kBlock), 0, h->stream, h->pool, (long)n * er::kUnitVox, h->dsum);
    ER_HIP_TRY(hipGetLastError());
  }
  ER_HIP_TRY(hipMemcpyAsync(sum, h->dsum, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

static int extract_points(er_tsdf_t h, float* out_host, long capacity, long* count, int surface);

int er_tsdf_extract_world(er_tsdf_t h, float* out_host, long capacity, long* count) { return extract_points(h, out_host, capacity, count, 0); }
int er_tsdf_extract_surface(er_tsdf_t h, float* out_host, long capacity, long* count) { return extract_points(h, out_host, capacity, count, 1); }

static int extract_points(er_tsdf_t h, float* out_host, long capacity, long* count, int surface) {
  if (!h || !count) return er::fail("er_tsdf_extract_world: NULL argument");
  ER_HIP_TRY(hipSetDevice(h->device));
  std::vector<int> keys, slots;
  if (sorted_units(h, keys, slots)) return 1;
  const int n = (int)keys.size();
  *count = 0;
  if (n == 0) return 0;
  const int nslab = n * 64;
  int *d_keys = nullptr, *d_slots = nullptr;
  long *d_cnt = nullptr, *d_off = nullptr;
  float4* d_out = nullptr;
  int rc = 0;
  std::vector<long> cnt((size_t)nslab), off((size_t)nslab);
  long total = 0;
#define ER_W(expr)                                                                              \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      rc = er::fail("er_tsdf_extract_world: %s failed: %s", #expr, hipGetErrorString(e_));      \
      goto done;                                                                                \
    }                                                                                           \
  } while (0)
  ER_W(hipMalloc((void**)&d_keys, (size_t)n * sizeof(int)));
  ER_W(hipMalloc((void**)&d_slots, (size_t)n * sizeof(int)));
  ER_W(hipMalloc((void**)&d_cnt, (size_t)nslab * sizeof(long)));
  ER_W(hipMalloc((void**)&d_off, (size_t)nslab * sizeof(long)));
  ER_W(hipMemcpyAsync(d_keys, keys.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
  ER_W(hipMemcpyAsync(d_slots, slots.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if (surface)
    hipLaunchKernelGGL(k_surface, dim3(nslab), dim3(64), 0, h->stream, h->pool, d_slots, d_keys, h->ht_key, h->ht_slot, h->ht_cap - 1, h->ht_shift,
                       d_cnt, d_off, (float4*)nullptr, 0);
  else
    hipLaunchKernelGGL(k_world, dim3(nslab), dim3(64), 0, h->stream, h->pool, d_slots, d_keys, d_cnt, d_off, (float4*)nullptr, 0);
  ER_W(hipGetLastError());
  ER_W(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)nslab * sizeof(long), hipMemcpyDeviceToHost, h->stream));
  ER_W(hipStreamSynchronize(h->stream));
  for (int s = 0; s < nslab; s++) {
    off[(size_t)s] = total;
    total += cnt[(size_t)s];
  }
  *count = total;
  if (out_host && total > 0) {
    if (capacity < total) {
      rc = er::fail("er_tsdf_extract_world: capacity %ld < %ld points", capacity, total);
      goto done;
    }
    ER_W(hipMalloc((void**)&d_out, (size_t)total * sizeof(float4)));
    ER_W(hipMemcpyAsync(d_off, off.data(), (size_t)nslab * sizeof(long), hipMemcpyHostToDevice, h->stream));
    if (surface)
      hipLaunchKernelGGL(k_surface, dim3(nslab), dim3(64), 0, h->stream, h->pool, d_slots, d_keys, h->ht_key, h->ht_slot, h->ht_cap - 1,
                         h->ht_shift, d_cnt, d_off, d_out, 1);
    else
      hipLaunchKernelGGL(k_world, dim3(nslab), dim3(64), 0, h->stream, h->pool, d_slots, d_keys, d_cnt, d_off, d_out, 1);
    ER_W(hipGetLastError());
    ER_W(hipMemcpyAsync(out_host, d_out, (size_t)total * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
    ER_W(hipStreamSynchronize(h->stream));
  }
#undef ER_W
done:
  if (d_keys) (void)hipFree(d_keys);
  if (d_slots) (void)hipFree(d_slots);
  if (d_cnt) (void)hipFree(d_cnt);
  if (d_off) (void)hipFree(d_off);
  if (d_out) (void)hipFree(d_out);
  return rc;
}

int er_mc_table(unsigned char out[256 * 16]) {
  if (!out) return er::fail("er_mc_table: NULL argument");
  memcpy(out, er::mc_table().tri, 256 * 16);
  return 0;
}

int er_tsdf_extract_mesh(er_tsdf_t h, float* tri_host, long capacity_triangles, long* n_triangles) {
  if (!h || !n_triangles) return er::fail("er_tsdf_extract_mesh: NULL argument");
  ER_HIP_TRY(hipSetDevice(h->device));
  std::vector<int> keys, slots;
  if (sorted_units(h, keys, slots)) return 1;
  const int n = (int)keys.size();
  *n_triangles = 0;
  if (n == 0) return 0;
  const int nslab = n * 64;
  int *d_keys = nullptr, *d_slots = nullptr;
  long *d_cnt = nullptr, *d_off = nullptr;
  unsigned char* d_tab = nullptr;
  float* d_out = nullptr;
  int rc = 0;
  std::vector<long> cnt((size_t)nslab), off((size_t)nslab);
  long total = 0;
#define ER_W(expr)                                                                              \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      rc = er::fail("er_tsdf_extract_mesh: %s failed: %s", #expr, hipGetErrorString(e_));       \
      goto done;                                                                                \
    }                                                                                           \
  } while (0)
  ER_W(hipMalloc((void**)&d_keys, (size_t)n * sizeof(int)));
  ER_W(hipMalloc((void**)&d_slots, (size_t)n * sizeof(int)));
  ER_W(hipMalloc((void**)&d_cnt, (size_t)nslab * sizeof(long)));
  ER_W(hipMalloc((void**)&d_off, (size_t)nslab * sizeof(long)));
  ER_W(hipMalloc((void**)&d_tab, 256 * 16));
  ER_W(hipMemcpyAsync(d_keys, keys.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
  ER_W(hipMemcpyAsync(d_slots, slots.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
  ER_W(hipMemcpyAsync(d_tab, er::mc_table().tri, 256 * 16, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_mesh, dim3(nslab), dim3(64), 0, h->stream, h->pool, d_slots, d_keys, h->ht_key, h->ht_slot, h->ht_cap - 1, h->ht_shift, d_tab,
                     d_cnt, d_off, (float*)nullptr, 0);
  ER_W(hipGetLastError());
  ER_W(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)nslab * sizeof(long), hipMemcpyDeviceToHost, h->stream));
  ER_W(hipStreamSynchronize(h->stream));
  for (int s = 0; s < nslab; s++) {
    off[(size_t)s] = total;
    total += cnt[(size_t)s];
  }
  *n_triangles = total;
  if (tri_host && total > 0) {
    if (capacity_triangles < total) {
      rc = er::fail("er_tsdf_extract_mesh: capacity %ld < %ld triangles", capacity_triangles, total);
      goto done;
    }
    ER_W(hipMalloc((void**)&d_out, (size_t)total * 9 * sizeof(float)));
    ER_W(hipMemcpyAsync(d_off, off.data(), (size_t)nslab * sizeof(long), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_mesh, dim3(nslab), dim3(64), 0, h->stream, h->pool, d_slots, d_keys, h->ht_key, h->ht_slot, h->ht_cap - 1, h->ht_shift, d_tab,
                       d_cnt, d_off, d_out, 1);
    ER_W(hipGetLastError());
    ER_W(hipMemcpyAsync(tri_host, d_out, (size_t)total * 9 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    ER_W(hipStreamSynchronize(h->stream));
  }
#undef ER_W
done:
  if (d_keys) (void)hipFree(d_keys);
  if (d_slots) (void)hipFree(d_slots);
  if (d_cnt) (void)hipFree(d_cnt);
  if (d_off) (void)hipFree(d_off);
  if (d_tab) (void)hipFree(d_tab);
  if (d_out) (void)hipFree(d_out);
  return rc;
}

static int export_units(er_tsdf_t h, const int* keys_host, int n_keys, float* dev_buf, int raw, const char* who) {
  if (!h || !keys_host || !dev_buf) return er::fail("%s: NULL argument", who);
  if (n_keys <= 0) return 0;
  ER_HIP_TRY(hipSetDevice(h->device));
  if (resolve_slots(h, keys_host, n_keys, false)) return 1;
  hipLaunchKernelGGL(k_export_weighted, dim3(er::kUnitVox / kBlock, n_keys), dim3(kBlock), 0, h->stream, h->pool,
                     h->slot_scratch, dev_buf, raw);
  ER_HIP_TRY(hipGetLastError());
  return 0;
}

static int import_units(er_tsdf_t h, const int* keys_host, int n_keys, const float* dev_buf, int raw, const char* who) {
  if (!h || !keys_host || !dev_buf) return er::fail("%s: NULL argument", who);
  if (n_keys <= 0) return 0;
  ER_HIP_TRY(hipSetDevice(h->device));
  undrop(h, keys_host, n_keys);
  if (resolve_slots(h, keys_host, n_keys, true)) return 1;
  hipLaunchKernelGGL(k_import_weighted, dim3(er::kUnitVox / kBlock, n_keys), dim3(kBlock), 0, h->stream, h->pool,
                     h->slot_scratch, dev_buf, raw);
  ER_HIP_TRY(hipGetLastError());
  return check_flags(h);
}

int er_tsdf_export_weighted(er_tsdf_t h, const int* keys_host, int n_keys, float* dev_buf) {
  return export_units(h, keys_host, n_keys, dev_buf, 0, "er_tsdf_export_weighted");
}
int er_tsdf_import_weighted(er_tsdf_t h, const int* keys_host, int n_keys, const float* dev_buf) {
  return import_units(h, keys_host, n_keys, dev_buf, 0, "er_tsdf_import_weighted");
}
int er_tsdf_export_raw(er_tsdf_t h, const int* keys_host, int n_keys, float* dev_buf) {
  return export_units(h, keys_host, n_keys, dev_buf, 1, "er_tsdf_export_raw");
}
int er_tsdf_import_raw(er_tsdf_t h, const int* keys_host, int n_keys, const float* dev_buf) {
  return import_units(h, keys_host, n_keys, dev_buf, 1, "er_tsdf_import_raw");
}

// ---- band records behind the C ABI (the device half of er_merge_protocol.h's OwnerMergeVolume) ------------------------------------------------
// chunk counts of the given units -> host: obs[n], band[n], wide[n]; the device copies stay in the band scratch ([counts n x 256 | wide n | offsets n]).
static int band_unit_counts(er_tsdf_t h, const int* keys_host, int n, std::vector<int>& obs, std::vector<int>& band, std::vector<int>& wide, const char* who) {
  if (resolve_slots(h, keys_host, n, false)) return 1;
  const size_t cnt_bytes = (size_t)n * 2 * kBandChunks * sizeof(int), wide_bytes = ((size_t)n * sizeof(int) + 15) & ~(size_t)15;
  if (ensure_band_scratch(h, cnt_bytes + wide_bytes + (size_t)n * sizeof(long) + 64)) return 1;
  int* d_cnt = (int*)h->band_scratch;
  int* d_wide = (int*)((char*)h->band_scratch + cnt_bytes);
  ER_HIP_TRY(hipMemsetAsync(d_wide, 0, (size_t)n * sizeof(int), h->stream));
  hipLaunchKernelGGL(k_band_count, dim3(kBandChunks / 4, n), dim3(256), 0, h->stream, h->pool, h->slot_scratch, d_cnt, d_wide);
  ER_HIP_TRY(hipGetLastError());
  std::vector<int> chunk((size_t)n * 2 * kBandChunks), slots((size_t)n);
  wide.assign((size_t)n, 0);
  ER_HIP_TRY(hipMemcpyAsync(chunk.data(), d_cnt, cnt_bytes, hipMemcpyDeviceToHost, h->stream));
  ER_HIP_TRY(hipMemcpyAsync(wide.data(), d_wide, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  ER_HIP_TRY(hipMemcpyAsync(slots.data(), h->slot_scratch, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  obs.assign((size_t)n, 0);
  band.assign((size_t)n, 0);
  for (int i = 0; i < n; i++) {
    if (slots[(size_t)i] < 0 || std::binary_search(h->dropped.begin(), h->dropped.end(), keys_host[i])) return er::fail("%s: this GPU holds no unit with key %d", who, keys_host[i]);
    for (int k = 0; k < kBandChunks; k++) {
      obs[(size_t)i] += chunk[(size_t)i * 2 * kBandChunks + k];
      band[(size_t)i] += chunk[(size_t)i * 2 * kBandChunks + kBandChunks + k];
    }
  }
  return 0;
}

int er_tsdf_band_sizes(er_tsdf_t h, const int* keys_host, int n, int* words_host) {
  if (!h || (n > 0 && (!keys_host || !words_host))) return er::fail("er_tsdf_band_sizes: NULL argument");
  if (n <= 0) return 0;
  ER_HIP_TRY(hipSetDevice(h->device));
  std::vector<int> obs, band, wide;
  if (band_unit_counts(h, keys_host, n, obs, band, wide, "er_tsdf_band_sizes")) return 1;
  for (int i = 0; i < n; i++) words_host[i] = (int)band_record_words(obs[(size_t)i], band[(size_t)i], wide[(size_t)i] & 1);
  return 0;
}

int er_tsdf_export_band(er_tsdf_t h, const int* keys_host, const int* words_host, int n, void* dev_block) {
  if (!h || (n > 0 && (!keys_host || !words_host || !dev_block))) return er::fail("er_tsdf_export_band: NULL argument");
  if (n <= 0) return 0;
  ER_HIP_TRY(hipSetDevice(h->device));
  std::vector<int> obs, band, wide;
  if (band_unit_counts(h, keys_host, n, obs, band, wide, "er_tsdf_export_band")) return 1;
  std::vector<long> off((size_t)n);
  long at = 0;
  for (int i = 0; i < n; i++) {
    const long w = band_record_words(obs[(size_t)i], band[(size_t)i], wide[(size_t)i] & 1);
    if (w != (long)words_host[i]) return er::fail("er_tsdf_export_band: the record of unit %d takes %ld words, the caller planned for %d (the volume changed since er_tsdf_band_sizes)", keys_host[i], w, words_host[i]);
    off[(size_t)i] = at;
    at += w;
  }
  const size_t cnt_bytes = (size_t)n * 2 * kBandChunks * sizeof(int), wide_bytes = ((size_t)n * sizeof(int) + 15) & ~(size_t)15;
  int* d_cnt = (int*)h->band_scratch;                            // (still holds the counts of exactly this key list)
  int* d_wide = (int*)((char*)h->band_scratch + cnt_bytes);
  long* d_off = (long*)((char*)h->band_scratch + cnt_bytes + wide_bytes);
  ER_HIP_TRY(hipMemcpyAsync(d_off, off.data(), (size_t)n * sizeof(long), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_band_pack, dim3(kBandChunks / 4, n), dim3(256), 0, h->stream, h->pool, h->slot_scratch, d_cnt, d_wide, d_off, (uint32_t*)dev_block);
  ER_HIP_TRY(hipGetLastError());
  ER_HIP_TRY(hipStreamSynchronize(h->stream));                   // (off is a host temporary; the block is complete when this returns)
  return 0;
}

int er_tsdf_merge_band(er_tsdf_t h, const int* keys_host, int n, const int* nsrc, const int* self_pos, const void* const* recs) {
  if (!h || (n > 0 && (!keys_host || !nsrc || !self_pos || !recs))) return er::fail("er_tsdf_merge_band: NULL argument");
  if (n <= 0) return 0;
  ER_HIP_TRY(hipSetDevice(h->device));
  if (resolve_slots(h, keys_host, n, false)) return 1;
  std::vector<int> slots((size_t)n);
  ER_HIP_TRY(hipMemcpyAsync(slots.data(), h->slot_scratch, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  std::vector<BandItem> items((size_t)n);
  for (int i = 0; i < n; i++) {
    if (slots[(size_t)i] < 0) return er::fail("er_tsdf_merge_band: the owner holds no unit with key %d", keys_host[i]);
    if (nsrc[i] < 0 || nsrc[i] > kBandMaxSrc || self_pos[i] < 0 || self_pos[i] > nsrc[i])
      return er::fail("er_tsdf_merge_band: unit %d has %d records (at most %d), own position %d", keys_host[i], nsrc[i], kBandMaxSrc, self_pos[i]);
    BandItem& b = items[(size_t)i];
    b.slot = slots[(size_t)i];
    b.nsrc = nsrc[i];
    b.self_pos = self_pos[i];
    b.pad = 0;
    for (int k = 0; k < kBandMaxSrc; k++) b.rec[k] = k < nsrc[i] ? (const uint32_t*)recs[(size_t)i * kBandMaxSrc + k] : nullptr;
  }
  if (ensure_band_scratch(h, items.size() * sizeof(BandItem))) return 1;
  ER_HIP_TRY(hipMemcpyAsync(h->band_scratch, items.data(), items.size() * sizeof(BandItem), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_band_merge, dim3(kBandChunks / 4, n), dim3(256), 0, h->stream, h->pool, (const BandItem*)h->band_scratch);
  ER_HIP_TRY(hipGetLastError());
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  return 0;
}

int er_tsdf_import_band(er_tsdf_t h, const int* keys_host, int n, const void* const* recs) {
  if (!h || (n > 0 && (!keys_host || !recs))) return er::fail("er_tsdf_import_band: NULL argument");
  if (n <= 0) return 0;
  ER_HIP_TRY(hipSetDevice(h->device));
  undrop(h, keys_host, n);
  if (resolve_slots(h, keys_host, n, true)) return 1;
  if (ensure_band_scratch(h, (size_t)n * sizeof(void*))) return 1;
  ER_HIP_TRY(hipMemcpyAsync(h->band_scratch, recs, (size_t)n * sizeof(void*), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_band_import, dim3(kBandChunks / 4, n), dim3(256), 0, h->stream, h->pool, h->slot_scratch, (const uint32_t* const*)h->band_scratch);
  ER_HIP_TRY(hipGetLastError());
  return check_flags(h);                                         // (synchronises: recs may be a host temporary)
}

int er_tsdf_drop_units(er_tsdf_t h, const int* keys_host, int n) {
  if (!h || (n > 0 && !keys_host)) return er::fail("er_tsdf_drop_units: NULL argument");
  if (n <= 0) return 0;
  ER_HIP_TRY(hipSetDevice(h->device));
  if (resolve_slots(h, keys_host, n, false)) return 1;
  hipLaunchKernelGGL(k_zero_units, dim3(kUnitVox / 2 / 256, n), dim3(256), 0, h->stream, h->pool, h->slot_scratch);
  ER_HIP_TRY(hipGetLastError());
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  h->dropped.insert(h->dropped.end(), keys_host, keys_host + n);
  std::sort(h->dropped.begin(), h->dropped.end());
  h->dropped.erase(std::unique(h->dropped.begin(), h->dropped.end()), h->dropped.end());
  return 0;
}

int er_tsdf_set_profiling(er_tsdf_t h, int enable) {
  if (!h) return er::fail("er_tsdf_set_profiling: NULL handle");
  ER_HIP_TRY(hipSetDevice(h->device));
  if (flush_resets(h) || drain_events(h)) return 1;         // (pending resets would add their unit visits after the counters are cleared)
  h->prof_stride = enable > 0 ? enable : 0;
  h->prof_tick = 0;
  h->ms_total = 0.0;
  h->launches = 0;
  h->frames_done = 0;
  ER_HIP_TRY(hipMemsetAsync(h->stats, 0, 4 * sizeof(unsigned long long), h->stream));
  return 0;
}

int er_tsdf_get_profile(er_tsdf_t h, double* integrate_ms_total, long* integrate_launches, long* frames,
                        long* unit_visits) {
  if (!h) return er::fail("er_tsdf_get_profile: NULL handle");
  ER_HIP_TRY(hipSetDevice(h->device));
  if (flush_resets(h)) return 1;
  ER_HIP_TRY(hipStreamSynchronize(h->stream));
  if (drain_events(h)) return 1;
  unsigned long long st[4] = {0, 0, 0, 0};
  ER_HIP_TRY(hipMemcpy(st, h->stats, sizeof st, hipMemcpyDeviceToHost));
  if (integrate_ms_total) *integrate_ms_total = h->ms_total;
  if (integrate_launches) *integrate_launches = h->launches;
  if (frames) *frames = h->frames_done;
  if (unit_visits) *unit_visits = (long)st[0];
  return 0;
}


}  // extern "C"

#endif  // ER_TSDF_TU == 0