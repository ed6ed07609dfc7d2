// GPU check of the exact arithmetic cores used by the voxel update (er_tsdf_math.h) against hipcc's IEEE
// operators, EXHAUSTIVELY where the domain is one float (sqrt, pixel rounding, the two constant divisions) and on 2^31 hashed operand
// triples per division scenario.  Test infrastructure: built by tests/conftest.py, run by tests/test_tsdf_gpu.py.
// Prints one line per check "name tested mismatches" and exits non-zero on any mismatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "er_tsdf_math.h"

using namespace er;

__device__ unsigned long long g_bad[8], g_cnt[8];
__device__ float g_ex[8][4][4];      // first few failing (n, d, got, want) per check
__device__ unsigned g_nex[8];

__device__ inline uint32_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (uint32_t)x;
}

// all float bit patterns
__global__ void k_sqrt_all() {
  unsigned long long bad = 0, cnt = 0;
  for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < (1ull << 31) + 1; b += (uint64_t)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((uint32_t)b);
    if (x != x) continue;
    const float got = sqrt_inrange(x), ref = sqrtf(x);
    if (x >= 0x1p-96f) {                       // in range (incl. +inf): same bits
      ++cnt;
      bad += __float_as_uint(got) != __float_as_uint(ref);
    } else {                                   // below: only "non-negative and tiny" is relied upon
      ++cnt;
      bad += !(got >= 0.0f && got < 1e-14f);
    }
  }
  atomicAdd(&g_bad[0], bad); atomicAdd(&g_cnt[0], cnt);
}

__global__ void k_pixel_all() {
  unsigned long long bad = 0, cnt = 0;
  for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < (1ull << 32); b += (uint64_t)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((uint32_t)b);
    for (int lim = 480; lim <= 640; lim += 160) {
      int p = -12345;
      const bool ok = pixel_index(x, (float)lim - 0.5f, p);
      const double r = floor((double)x + 0.5);
      const bool ok_ref = (r >= 0.0) & (r < (double)lim);
      ++cnt;
      bad += (ok != ok_ref) || (ok && p != (int)r);
    }
  }
  atomicAdd(&g_bad[1], bad); atomicAdd(&g_cnt[1], cnt);
}

// band_quotient_core vs the float64 '/' for every float in [-0.03f, 0.03f] except -0 (unreachable, see er_tsdf_math.h);
// div1000_core vs '/' for +0 and every float >= 1 including +inf (NaN in, NaN out).
__global__ void k_const_div_all() {
  unsigned long long bad5 = 0, cnt5 = 0, bad6 = 0, cnt6 = 0;
  const uint32_t top = __float_as_uint(0.03f);
  for (uint64_t b = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b < (1ull << 32); b += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t u = (uint32_t)b;
    const float x = __uint_as_float(u);
    if ((u & 0x7fffffffu) <= top && u != 0x80000000u) {
      const double got = band_quotient_core(x), ref = (double)x / kTsdfTrunc;
      ++cnt5;
      bad5 += __double_as_longlong(got) != __double_as_longlong(ref);
    }
    if (u == 0u || (u >= 0x3f800000u && u <= 0x7fffffffu)) {
      const float got = div1000_core(x), ref = x / 1000.f;
      ++cnt6;
      bad6 += (ref != ref) ? !(got != got) : (__float_as_uint(got) != __float_as_uint(ref));
    }
  }
  atomicAdd(&g_bad[5], bad5); atomicAdd(&g_cnt[5], cnt5);
  atomicAdd(&g_bad[6], bad6); atomicAdd(&g_cnt[6], cnt6);
}

// Hashed operand triples.
//   mode 0 "projection": depth d in [2^-30, 2^30], numerators over the WHOLE float range (incl. denormals, 0).
//           Inside the v_div_scale-free domain the bits must match (a -0 numerator gives +0 instead of -0); outside, only what voxel_update relies on:
//           the pixel decision of pixel_index(q + cx) for cx in {0, 0.3, 319.5}.
//   mode 1 "weight update": d = integer 1..2^24, numerators 2^-60..2^30 and exact small fractions: bits must match.
//   mode 2 "camera-like": d in [1/16, 16], numerators 2^-12..2^12: bits must match.
__global__ void k_div(int mode, uint64_t n_samples) {
  unsigned long long bad = 0, cnt = 0;
  for (uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; s < n_samples; s += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t a = mix(s * 2 + 0x1234567), b = mix(s * 2 + 0x89abcdef), c = mix(s * 3 + 77);
    auto mk = [](uint32_t h, int spread) {
      const int e = 127 - spread + (int)((h >> 23) % (2 * spread + 1));
      return __uint_as_float((h & 0x807fffffu) | ((uint32_t)e << 23));
    };
    float n0, n1, d;
    if (mode == 0) {
      d = fabsf(mk(b, 30));
      n0 = __uint_as_float(a);                                  // any bit pattern
      n1 = __uint_as_float(c);
      if (n0 != n0 || fabsf(n0) == INFINITY) n0 = 0.0f;
      if (n1 != n1 || fabsf(n1) == INFINITY) n1 = -0.0f;
    } else if (mode == 1) { d = (float)(1 + (b & 0xffffff)); n0 = mk(a, 30) * ((c & 1) ? 1.0f : 0x1p-30f); n1 = (float)(int)(a >> 8) * 0x1p-20f; }
    else { d = fabsf(mk(b, 4)); n0 = mk(a, 12); n1 = mk(c, 12); }
    float q0, q1;
    div2_inrange(n0, n1, d, q0, q1);
    const float r0 = n0 / d, r1 = n1 / d, s0 = div_inrange(n0, d);
    const float q[3] = {q0, q1, s0}, r[3] = {r0, r1, r0}, n[3] = {n0, n1, n0};
    for (int i = 0; i < 3; ++i) {
      ++cnt;
      const float an = fabsf(n[i]);
      const bool domain = mode != 0 || (an >= 0x1p-100f && an < d * 0x1p95f && an >= d * 0x1p-125f);
      if (domain) {
        const bool ne = __float_as_uint(q[i]) != __float_as_uint(r[i]);
        bad += ne;
        if (ne) {
          const unsigned k = atomicAdd(&g_nex[2 + mode], 1u);
          if (k < 4) { g_ex[2 + mode][k][0] = n[i]; g_ex[2 + mode][k][1] = d; g_ex[2 + mode][k][2] = q[i]; g_ex[2 + mode][k][3] = r[i]; }
        }
      } else {
        const float cxs[3] = {0.0f, 0.3f, 319.5f};
        for (int k = 0; k < 3; ++k) {
          int pq = -1, pr = -1;
          const bool oq = pixel_index(q[i] + cxs[k], 639.5f, pq), orf = pixel_index(r[i] + cxs[k], 639.5f, pr);
          bad += (oq != orf) || (oq && pq != pr);
        }
      }
    }
  }
  atomicAdd(&g_bad[2 + mode], bad); atomicAdd(&g_cnt[2 + mode], cnt);
}

int main() {
  unsigned long long zero[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bad), zero, sizeof zero);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cnt), zero, sizeof zero);
  unsigned zu[8] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_nex), zu, sizeof zu);
  hipLaunchKernelGGL(k_sqrt_all, dim3(4096), dim3(256), 0, 0);
  hipLaunchKernelGGL(k_pixel_all, dim3(4096), dim3(256), 0, 0);
  hipLaunchKernelGGL(k_const_div_all, dim3(4096), dim3(256), 0, 0);
  for (int m = 0; m < 3; ++m) hipLaunchKernelGGL(k_div, dim3(4096), dim3(256), 0, 0, m, 1ull << 31);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failure\n"); return 2; }
  unsigned long long bad[8], cnt[8];
  (void)hipMemcpyFromSymbol(bad, HIP_SYMBOL(g_bad), sizeof bad);
  (void)hipMemcpyFromSymbol(cnt, HIP_SYMBOL(g_cnt), sizeof cnt);
  float ex[8][4][4]; unsigned nex[8];
  (void)hipMemcpyFromSymbol(ex, HIP_SYMBOL(g_ex), sizeof ex);
  (void)hipMemcpyFromSymbol(nex, HIP_SYMBOL(g_nex), sizeof nex);
  const char* names[7] = {"sqrt_inrange", "pixel_index", "div_projection", "div_weight", "div_camera", "band_quotient", "div1000"};
  int rc = 0;
  for (int i = 0; i < 7; ++i) {
    printf("%s tested %llu mismatches %llu\n", names[i], cnt[i], bad[i]);
    for (unsigned k = 0; k < nex[i] && k < 4; ++k)
      fprintf(stderr, "  %s: n=%a d=%a core=%a operator=%a\n", names[i], ex[i][k][0], ex[i][k][1], ex[i][k][2], ex[i][k][3]);
    if (bad[i] || !cnt[i]) rc = 1;
  }
  return rc;
}
