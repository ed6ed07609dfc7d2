// er_common.h -- shared host-side helpers of liber_hip.so (error reporting, HIP call checking,
// small row-major 4x4 double algebra).  No torch, no Eigen: the system has no linear-algebra
// library outside /root/reference, and the C ABI must stay plain C.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace er {

// Thread-local error text behind er_last_error().
char* error_buffer();
int fail(const char* fmt, ...);

#define ER_HIP_TRY(expr)                                                                          \
  do {                                                                                            \
    hipError_t er_e_ = (expr);                                                                    \
    if (er_e_ != hipSuccess)                                                                      \
      return ::er::fail("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(er_e_)); \
  } while (0)

// C = A * B, row-major 4x4 float64, each coefficient summed as ((a0*b0 + a1*b1) + a2*b2) + a3*b3
// (the order Eigen 3.1.2's coefficient-based product uses; see oracle/tsdf_oracle.c).
inline void mat4_mul(const double* A, const double* B, double* C) {
  double t[16];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++)
      t[r * 4 + c] = ((A[r * 4 + 0] * B[0 * 4 + c] + A[r * 4 + 1] * B[1 * 4 + c]) + A[r * 4 + 2] * B[2 * 4 + c]) +
                     A[r * 4 + 3] * B[3 * 4 + c];
  memcpy(C, t, sizeof t);
}

// General 4x4 inverse (cofactor expansion, float64).  Returns false if singular.
bool mat4_inverse(const double* m, double* out);

}  // namespace er
