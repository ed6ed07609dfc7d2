// d2h_paths.hip -- how fast do 83 MB (50 lists of 1.66 MB) get from HBM into page-locked host memory on this box?
//   (a) a kernel that stores straight into the host buffer (what k_compact does for in-place lists), 16-byte and 8-byte stores;
//   (b) hipMemcpyAsync of the 50 pieces on 1 / 2 / 4 streams (what er_find_correspondence_batch did until round 6).
// build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/d2h_paths.hip -o scripts/ubench/d2h_paths ; run: scripts/ubench/d2h_paths
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_store16(const int4* __restrict__ src, int4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_store8(const int2* __restrict__ src, int2* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
  const size_t piece = 1664000, np = 50, bytes = piece * np;
  char *d = nullptr, *h = nullptr;
  CK(hipMalloc((void**)&d, bytes));
  CK(hipHostMalloc((void**)&h, bytes, hipHostMallocDefault));
  CK(hipMemset(d, 1, bytes));
  hipStream_t st[4];
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  for (int rep = 0; rep < 3; rep++) {
    for (int grid : {256, 1024, 4096, 16384}) {
      CK(hipDeviceSynchronize());
      auto t0 = now();
      hipLaunchKernelGGL(k_store16, dim3(grid), dim3(256), 0, st[0], (const int4*)d, (int4*)h, bytes / 16);
      CK(hipStreamSynchronize(st[0]));
      const double a = ms(t0, now());
      t0 = now();
      hipLaunchKernelGGL(k_store8, dim3(grid), dim3(256), 0, st[0], (const int2*)d, (int2*)h, bytes / 8);
      CK(hipStreamSynchronize(st[0]));
      const double b = ms(t0, now());
      printf("rep %d kernel stores, grid %5d: 16-byte %.2f ms (%.1f GB/s), 8-byte %.2f ms (%.1f GB/s)\n", rep, grid, a, bytes / a / 1e6, b, bytes / b / 1e6);
    }
    for (int ns : {1, 2, 4}) {
      CK(hipDeviceSynchronize());
      auto t0 = now();
      for (size_t p = 0; p < np; p++) CK(hipMemcpyAsync(h + p * piece, d + p * piece, piece, hipMemcpyDeviceToHost, st[p % ns]));
      for (int s = 0; s < ns; s++) CK(hipStreamSynchronize(st[s]));
      const double a = ms(t0, now());
      printf("rep %d hipMemcpyAsync x %zu pieces on %d stream(s): %.2f ms (%.1f GB/s)\n", rep, np, ns, a, bytes / a / 1e6);
    }
    {
      CK(hipDeviceSynchronize());
      auto t0 = now();
      CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st[0]));
      CK(hipStreamSynchronize(st[0]));
      const double a = ms(t0, now());
      printf("rep %d hipMemcpyAsync, ONE piece of %zu MB: %.2f ms (%.1f GB/s)\n", rep, bytes >> 20, a, bytes / a / 1e6);
    }
  }
  return 0;
}
