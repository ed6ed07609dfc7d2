// What a candidate load of path B's search costs the vector L1 / the LDS: wave64 gathers of 4 / 8 / 12 / 16 bytes per lane from a table that stays
// in L1 (16 KB), with 64 / 8 / 1 distinct addresses per instruction, against ds_read_b128 with the same address patterns.
// Build + run:  hipcc --offload-arch=gfx950 -O3 scripts/ubench/gather_rates.hip -o /tmp/gather_rates && /tmp/gather_rates
// Output: CU clocks per wave instruction at 8 waves per SIMD (the issue limit is 1 per 4 clocks per SIMD = 1 per clock per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kEntries = 1024;          // 16 KB of float4
constexpr int kIters = 2048;

template <int BYTES, int GROUP>
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ tab, float* __restrict__ out, unsigned long long* __restrict__ clk) {
  const int lane = threadIdx.x & 63;
  unsigned idx = ((threadIdx.x / GROUP) * 2654435761u + blockIdx.x * 97u) % kEntries;      // GROUP lanes share an entry
  float acc = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < kIters; i++) {
    const char* p = (const char*)tab + (size_t)idx * 16;
    if (BYTES == 16) { f4 v = *(const f4*)p; acc += v.x + v.w; }
    if (BYTES == 12) { f3 v = *(const f3*)p; acc += v.x + v.z; }
    if (BYTES == 8) { f2 v = *(const f2*)p; acc += v.x + v.y; }
    if (BYTES == 4) { acc += *(const float*)p; }
    idx = (idx * 5u + 1u + (__float_as_uint(acc) & 0u)) % kEntries;          // next entry: depends on the loaded value only formally (no serialisation beyond the wait)
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
  (void)lane;
}

template <int GROUP>
__global__ __launch_bounds__(256) void k_lds(const float* __restrict__ tab, float* __restrict__ out, unsigned long long* __restrict__ clk) {
  __shared__ f4 s[kEntries];
  for (int i = threadIdx.x; i < kEntries; i += 256) s[i] = ((const f4*)tab)[i];
  __syncthreads();
  unsigned idx = ((threadIdx.x / GROUP) * 2654435761u + blockIdx.x * 97u) % kEntries;
  float acc = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < kIters; i++) {
    f4 v = s[idx];
    acc += v.x + v.w;
    idx = (idx * 5u + 1u + (__float_as_uint(acc) & 0u)) % kEntries;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <class K>
static void run(const char* name, K kernel, const float* tab, float* out, unsigned long long* clk, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, tab, out, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, tab, out, clk);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // blocks = 8 per CU x 256 CUs: every CU runs 8 x 4 waves = 8 per SIMD; wave instructions per CU = 32 waves x kIters
  const double cu_clocks = ms * 1e-3 * 2.4e9;
  printf("%-44s %8.3f ms  -> %6.2f CU clocks per wave instruction (at 2.4 GHz)\n", name, ms, cu_clocks / (32.0 * kIters));
}

int main() {
  float *tab, *out;
  unsigned long long* clk;
  const int blocks = 256 * 8;
  hipMalloc(&tab, kEntries * 16 + 64);
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMalloc(&clk, blocks * 8);
  std::vector<float> h(kEntries * 4 + 16, 1.0f);
  hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run("global 16 B/lane, 64 distinct entries", k_gather<16, 1>, tab, out, clk, blocks);
  run("global 12 B/lane, 64 distinct entries", k_gather<12, 1>, tab, out, clk, blocks);
  run("global  8 B/lane, 64 distinct entries", k_gather<8, 1>, tab, out, clk, blocks);
  run("global  4 B/lane, 64 distinct entries", k_gather<4, 1>, tab, out, clk, blocks);
  run("global 16 B/lane,  8 distinct (groups of 8)", k_gather<16, 8>, tab, out, clk, blocks);
  run("global 16 B/lane,  1 distinct (wave-uniform)", k_gather<16, 64>, tab, out, clk, blocks);
  run("global  4 B/lane,  8 distinct (groups of 8)", k_gather<4, 8>, tab, out, clk, blocks);
  run("LDS    16 B/lane, 64 distinct entries", k_lds<1>, tab, out, clk, blocks);
  run("LDS    16 B/lane,  8 distinct (groups of 8)", k_lds<8>, tab, out, clk, blocks);
  run("LDS    16 B/lane,  1 distinct (wave-uniform)", k_lds<64>, tab, out, clk, blocks);
  return 0;
}
