// er_common.cpp -- error reporting and small host-side algebra of liber_hip.so.
#include "er_common.h"

#include "../../include/er_hip.h"

#include <cstdlib>

// HIP multiplexes a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams that share a queue
// run in order.  A TSDF volume drives four streams (voxel pass, two pre-passes, host-frame copies) next to the caller's own,
// and their overlap is the point (er_tsdf.hip: run_batch).  The variable is read when the HIP runtime initialises, i.e. at the
// process's first HIP call: setting it here -- never overriding the user's value -- covers every program that has not touched
// HIP before it loads this library; programs that have (a Python process that used torch.cuda first) set it themselves
// (bench.py, elasticreconstruction_amd/__init__.py).
__attribute__((constructor)) static void er_request_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

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
