// er_common.cpp -- error reporting and small host-side algebra of liber_hip.so.
#include "er_common.h"

#include "../../include/er_hip.h"

namespace er {

char* error_buffer() {
  static thread_local char buf[1024] = {0};
  return buf;
}

int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 1024, fmt, ap);
  va_end(ap);
  return 1;
}

}  // namespace er

extern "C" {

const char* er_last_error(void) { return er::error_buffer(); }

int er_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int er_abi_version(void) { return 2; }

void* er_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    er::fail("er_host_alloc: hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(hipGetLastError()));
    return nullptr;
  }
  return p;
}

int er_host_copy_h2d(void* dev_dst, const void* host_src, size_t bytes) {
  ER_HIP_TRY(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
  return 0;
}

int er_host_free(void* p) {
  if (p) ER_HIP_TRY(hipHostFree(p));
  return 0;
}

}  // extern "C"
