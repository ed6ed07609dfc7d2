// Which XCD does a workgroup run on?  s_getreg_b32 HW_REG_XCC_ID against blockIdx.x % 8, for a persistent grid of 2 x 256 workgroups of 256 threads.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 2) void k(int* out, int spin) {
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = x;
  // keep the workgroup resident for a while so that the whole grid is co-resident (as k_integrate's is)
  float a = (float)threadIdx.x;
  for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;
  if (a == 12345.f) out[0] = -1;
}
int main() {
  const int n = 512;
  int* d; hipMalloc(&d, n * sizeof(int));
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d, 200000);
  int h[n]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int hist[16][8] = {}, raw_max = 0, agree = 0;
  for (int b = 0; b < n; b++) { raw_max = h[b] > raw_max ? h[b] : raw_max; hist[h[b] & 15][b % 8]++; agree += (h[b] & 7) == b % 8; }
  printf("raw XCC_ID register max 0x%x; (id & 7) == blockIdx %% 8 for %d of %d workgroups\n", raw_max, agree, n);
  for (int x = 0; x < 16; x++) { int s = 0; for (int m = 0; m < 8; m++) s += hist[x][m]; if (s) { printf("id %2d: %3d workgroups; by blockIdx %% 8:", x, s); for (int m = 0; m < 8; m++) printf(" %3d", hist[x][m]); printf("\n"); } }
  return 0;
}
