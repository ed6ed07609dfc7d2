// Issue-rate microbenchmark for the VALU instructions the voxel update is made of (MI355X, gfx950).
// Each kernel runs ITER x 32 independent instances of one instruction per wave; the grid fills every SIMD with
// `waves` waves.  Output: cycles per wave-instruction per SIMD (4 = full rate for a 64-wide wave on 16 lanes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define ITER 2000
#define REP8(x) x x x x x x x x

#define KERNEL(name, decl, body)                                             \
  __global__ void name(float* out, float seed) {                               \
    decl;                                                                      \
    for (int it = 0; it < ITER; ++it) { REP8(body) REP8(body) REP8(body) REP8(body) }  \
    if (seed == 12345.f) out[threadIdx.x] = (float)a0 + (float)a1 + (float)a2 + (float)a3; \
  }

#define F32DECL float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b = seed * 0.5f
#define F64DECL double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, b = seed * 0.5
// four independent chains; one "body" = 4 instructions -> 128 per iteration
#define OP1(ins) asm volatile(ins " %0, %0\n" ins " %1, %1\n" ins " %2, %2\n" ins " %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
#define OP2(ins) asm volatile(ins " %0, %0, %4\n" ins " %1, %1, %4\n" ins " %2, %2, %4\n" ins " %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
#define OP3(ins) asm volatile(ins " %0, %0, %4, %4\n" ins " %1, %1, %4, %4\n" ins " %2, %2, %4, %4\n" ins " %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
#define CMP(ins) asm volatile(ins " vcc, %0, %4\n" ins " vcc, %1, %4\n" ins " vcc, %2, %4\n" ins " vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");

KERNEL(k_add_f32, F32DECL, OP2("v_add_f32"))
KERNEL(k_fma_f32, F32DECL, OP3("v_fma_f32"))
KERNEL(k_rcp_f32, F32DECL, OP1("v_rcp_f32"))
KERNEL(k_sqrt_f32, F32DECL, OP1("v_sqrt_f32"))
KERNEL(k_floor_f32, F32DECL, OP1("v_floor_f32"))
KERNEL(k_cvt_i32_f32, F32DECL, OP1("v_cvt_i32_f32"))
KERNEL(k_cmp_f32, F32DECL, CMP("v_cmp_lt_f32"))
KERNEL(k_divfixup_f32, F32DECL, OP3("v_div_fixup_f32"))
KERNEL(k_divfmas_f32, F32DECL, OP3("v_div_fmas_f32"))
KERNEL(k_mul_f32, F32DECL, OP2("v_mul_f32"))
KERNEL(k_sub_f32, F32DECL, OP2("v_sub_f32"))
KERNEL(k_max_f32, F32DECL, OP2("v_max_f32"))
KERNEL(k_fmac_f32, F32DECL, OP2("v_fmac_f32"))
KERNEL(k_mov_b32, F32DECL, OP1("v_mov_b32"))
KERNEL(k_cvt_f32_i32, F32DECL, OP1("v_cvt_f32_i32"))
// (round 2) integer and select instructions of the address / pixel-index arithmetic
#define I32DECL unsigned a0 = (unsigned)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = a0 * 3 + 1
KERNEL(k_add_u32, I32DECL, OP2("v_add_u32"))
KERNEL(k_and_b32, I32DECL, OP2("v_and_b32"))
KERNEL(k_mul_lo_u32, I32DECL, OP2("v_mul_lo_u32"))
KERNEL(k_mul_u32_u24, I32DECL, OP2("v_mul_u32_u24"))
KERNEL(k_mad_u32_u24, I32DECL, OP3("v_mad_u32_u24"))
KERNEL(k_lshl_add_u32, I32DECL, OP3("v_lshl_add_u32"))
KERNEL(k_cndmask_b32, I32DECL, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\nv_cndmask_b32 %1, %1, %4, vcc\nv_cndmask_b32 %2, %2, %4, vcc\nv_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");)
KERNEL(k_add_f64, F64DECL, OP2("v_add_f64"))
KERNEL(k_fma_f64, F64DECL, OP3("v_fma_f64"))
KERNEL(k_mul_f64, F64DECL, OP2("v_mul_f64"))
KERNEL(k_floor_f64, F64DECL, OP1("v_floor_f64"))
KERNEL(k_rcp_f64, F64DECL, OP1("v_rcp_f64"))
KERNEL(k_cmp_f64, F64DECL, CMP("v_cmp_lt_f64"))

__global__ void k_cvt_f64_f32(float* out, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3; double d0, d1, d2, d3;
  for (int it = 0; it < ITER; ++it) {
#define B asm volatile("v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %5\nv_cvt_f64_f32 %2, %6\nv_cvt_f64_f32 %3, %7" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    REP8(B) REP8(B) REP8(B) REP8(B)
  }
  if (seed == 12345.f) out[threadIdx.x] = (float)(d0 + d1 + d2 + d3);
}
#define PKKERNEL(name, ins)                                                                                          \
  __global__ void name(float* out, float seed) {                                                                     \
    typedef float f2 __attribute__((ext_vector_type(2)));                                                            \
    f2 a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b = a0 * .5f;                                 \
    for (int it = 0; it < ITER; ++it) {                                                                               \
      REP8(asm volatile(ins " %0, %0, %4\n" ins " %1, %1, %4\n" ins " %2, %2, %4\n" ins " %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) \
      REP8(asm volatile(ins " %0, %0, %4\n" ins " %1, %1, %4\n" ins " %2, %2, %4\n" ins " %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) \
      REP8(asm volatile(ins " %0, %0, %4\n" ins " %1, %1, %4\n" ins " %2, %2, %4\n" ins " %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) \
      REP8(asm volatile(ins " %0, %0, %4\n" ins " %1, %1, %4\n" ins " %2, %2, %4\n" ins " %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) \
    }                                                                                                                 \
    if (seed == 12345.f) out[threadIdx.x] = a0.x + a1.y + a2.x + a3.y;                                               \
  }
PKKERNEL(k_pk_mul_f32, "v_pk_mul_f32")
PKKERNEL(k_pk_add_f32, "v_pk_add_f32")
__global__ void k_lshl_add_u64(float* out, float seed) {
  unsigned long long a0 = (unsigned long long)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = a0 * 3 + 1;
  for (int it = 0; it < ITER; ++it) {
#define L64 asm volatile("v_lshl_add_u64 %0, %0, 2, %4\nv_lshl_add_u64 %1, %1, 2, %4\nv_lshl_add_u64 %2, %2, 2, %4\nv_lshl_add_u64 %3, %3, 2, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    REP8(L64) REP8(L64) REP8(L64) REP8(L64)
  }
  if (seed == 12345.f) out[threadIdx.x] = (float)(a0 + a1 + a2 + a3);
}
__global__ void k_pk_fma_f32(float* out, float seed) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b = a0 * .5f;
  for (int it = 0; it < ITER; ++it) {
#define P asm volatile("v_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    REP8(P) REP8(P) REP8(P) REP8(P)
  }
  if (seed == 12345.f) out[threadIdx.x] = a0.x + a1.y + a2.x + a3.y;
}

template <typename K>
void run(const char* name, K k, float* out, int waves_per_simd) {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, block = 256;                 // 4 waves per block = one per SIMD
  const int grid = cus * waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, out, 1.0f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, out, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double clk = p.clockRate * 1e3;                                 // Hz
  const double insts_per_simd = (double)ITER * 128 * waves_per_simd;
  printf("%-16s waves/SIMD %d  %.3f ms  %.2f cycles per wave-instruction (at %.0f MHz)\n", name, waves_per_simd, ms,
         ms * 1e-3 * clk / insts_per_simd, clk / 1e6);
}

int main() {
  float* out; hipMalloc(&out, 4096);
  for (int w : {2, 8}) {
    run("v_add_f32", k_add_f32, out, w); run("v_fma_f32", k_fma_f32, out, w); run("v_pk_fma_f32", k_pk_fma_f32, out, w);
    run("v_rcp_f32", k_rcp_f32, out, w); run("v_sqrt_f32", k_sqrt_f32, out, w);
    run("v_floor_f32", k_floor_f32, out, w); run("v_cvt_i32_f32", k_cvt_i32_f32, out, w); run("v_cmp_lt_f32", k_cmp_f32, out, w);
    run("v_div_fixup_f32", k_divfixup_f32, out, w); run("v_div_fmas_f32", k_divfmas_f32, out, w);
    run("v_add_f64", k_add_f64, out, w); run("v_mul_f64", k_mul_f64, out, w); run("v_fma_f64", k_fma_f64, out, w);
    run("v_floor_f64", k_floor_f64, out, w); run("v_rcp_f64", k_rcp_f64, out, w); run("v_cmp_lt_f64", k_cmp_f64, out, w);
    run("v_cvt_f64_f32", k_cvt_f64_f32, out, w);
    run("v_mul_f32", k_mul_f32, out, w); run("v_sub_f32", k_sub_f32, out, w); run("v_max_f32", k_max_f32, out, w); run("v_fmac_f32", k_fmac_f32, out, w);
    run("v_mov_b32", k_mov_b32, out, w); run("v_cvt_f32_i32", k_cvt_f32_i32, out, w);
    run("v_pk_mul_f32", k_pk_mul_f32, out, w); run("v_pk_add_f32", k_pk_add_f32, out, w);
    run("v_add_u32", k_add_u32, out, w); run("v_and_b32", k_and_b32, out, w); run("v_mul_lo_u32", k_mul_lo_u32, out, w);
    run("v_mul_u32_u24", k_mul_u32_u24, out, w); run("v_mad_u32_u24", k_mad_u32_u24, out, w); run("v_lshl_add_u32", k_lshl_add_u32, out, w);
    run("v_cndmask_b32", k_cndmask_b32, out, w); run("v_lshl_add_u64", k_lshl_add_u64, out, w);
  }
  return 0;
}
