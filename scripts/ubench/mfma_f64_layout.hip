// Which (lane, register) holds D[i][j] of v_mfma_f64_16x16x4_f64 on gfx950?  One wave computes D = A(16x4) * B(4x16) with
// A[i][k] = lane i + 16 k's a-operand, B[k][j] = lane j + 16 k's b-operand (the documented input layout), and the host
// tests the two candidate output layouts against a CPU product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void k(const double* a, const double* b, double* d) {
  double4_t acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  for (int v = 0; v < 4; v++) d[threadIdx.x * 4 + v] = acc[v];
}
int main() {
  double ha[64], hb[64], hd[256], *a, *b, *d;
  for (int l = 0; l < 64; l++) { ha[l] = 1.0 + 0.37 * l + 0.01 * l * l; hb[l] = 2.0 - 0.11 * l + 0.003 * l * l; }
  hipMalloc(&a, sizeof ha); hipMalloc(&b, sizeof hb); hipMalloc(&d, sizeof hd);
  hipMemcpy(a, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof hb, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d);
  hipMemcpy(hd, d, sizeof hd, hipMemcpyDeviceToHost);
  double ref[16][16];
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int kk = 0; kk < 4; kk++) s += ha[i + 16 * kk] * hb[j + 16 * kk]; ref[i][j] = s; }
  int okA = 1, okB = 1;
  for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) {
    const double got = hd[l * 4 + v];
    const double wa = ref[4 * (l / 16) + v][l % 16], wb = ref[(l / 16) + 4 * v][l % 16];
    if (fabs(got - wa) > 1e-9 * fabs(wa)) okA = 0;
    if (fabs(got - wb) > 1e-9 * fabs(wb)) okB = 0;
  }
  printf("layout i=4*(lane/16)+v, j=lane%%16: %s\nlayout i=(lane/16)+4*v, j=lane%%16: %s\n", okA ? "MATCH" : "no", okB ? "MATCH" : "no");
  return 0;
}
