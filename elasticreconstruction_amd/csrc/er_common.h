// er_common.h -- shared host-side helpers of liber_hip.so (error reporting, HIP call checking,
// small row-major 4x4 double algebra).  No torch, no Eigen: the system has no linear-algebra
// library outside /root/reference, and the C ABI must stay plain C.
#pragma once

#include <hip/hip_runtime.h>

#include "er_mat4.h"

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

}  // namespace er

// internal accessors of a TSDF handle for the other translation units of the library (not part of the C ABI)
struct er_tsdf_s;
namespace er {
hipStream_t tsdf_stream(er_tsdf_s* h);
int tsdf_device(er_tsdf_s* h);
}  // namespace er
