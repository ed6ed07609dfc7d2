// er_common.cpp -- error reporting and small host-side algebra of liber_hip.so.
#include "er_common.h"

#include "../../include/er_hip.h"

#include <cstdlib>

// HIP multiplexes a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams that share a queue
// run in order.  A TSDF volume drives four streams (voxel pass, two pre-passes, host-frame copies) next to the caller's own,
// and their overlap is the point (er_tsdf.hip: run_batch; measured 125.5 k frames/s with 8 queues, 106.0 k with 4).  The
// variable is read when the HIP runtime initialises, i.e. at the process's first HIP call, and it changes the queue set-up of
// EVERY GPU user of the process -- so the library does not touch it behind the host's back: a host opts in by calling
// er_request_hw_queues() before its first HIP call (bin/Integrate, bin/BuildCorrespondence and bench.py do), or exports the
// variable itself.
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

int er_abi_version(void) { return 3; }

int er_request_hw_queues(int n) {
  char v[16];
  snprintf(v, sizeof v, "%d", n > 0 ? n : 8);
  setenv("GPU_MAX_HW_QUEUES", v, 0);                      // never overrides the user's value
  const char* now = getenv("GPU_MAX_HW_QUEUES");
  return now ? atoi(now) : 0;
}

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

void* er_device_alloc(size_t bytes, int device) {
  void* p = nullptr;
  if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
    er::fail("er_device_alloc: hipMalloc(%zu) on device %d failed: %s", bytes, device, hipGetErrorString(hipGetLastError()));
    return nullptr;
  }
  return p;
}

int er_device_free(void* p) {
  if (p) ER_HIP_TRY(hipFree(p));
  return 0;
}

int er_device_copy_d2h(void* host_dst, const void* dev_src, size_t bytes) {
  ER_HIP_TRY(hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
