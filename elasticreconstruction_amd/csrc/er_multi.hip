// er_multi.hip -- the multi-GPU side of liber_hip.so: the ONE collective of the pipeline, issued from the
// library itself so that the C++ drop-in programs (and any other host) need neither torch nor MPI.
//
//   frames (path A, as BASELINE.json prescribes): contiguous frame blocks per GPU into private volumes, then
//       er_tsdf_allreduce = [agree on the key count: all-reduce(MAX) of one int] -> [all-gather of the touched unit keys,
//       padded to that count] -> union + who touched what -> ONE ncclReduce / ncclAllReduce (sum, float) over the
//       [key][sdf*weight | weight] planes of the units two or more ranks touched -> import (weight = W, sdf = SW / W), and one
//       grouped ncclSend / ncclRecv step for the units only one rank touched (raw, bit for bit; er_merge_protocol.h).  The running
//       mean with unit weights is a sum (TSDFVolume.cpp:93-94: sdf' = (sdf w + tsdf) / (w + 1), w' = w + 1), so this equals the
//       sequential result up to the float32 rounding order: weights exact, sdf within 1e-5 (exact in single-toucher units).
//   units  (the bit-exact alternative): er_tsdf_set_unit_shard in er_tsdf.hip -- no collective at all.
//   pairs  (path B): independent, no collective (BuildCorrespondence --gpus).
//
// RCCL (= NCCL's API on ROCm; xGMI between the GPUs of a node) is loaded on first use with dlopen, by soname: a process that
// already carries an RCCL (PyTorch bundles one) keeps that single instance; the C++ programs pick up /opt/rocm/lib/librccl.so.1.
// One communicator per GPU: er_comm_create (one process per GPU; the 128-byte id travels out of band -- a file, a socket,
// torch.distributed) or er_comm_create_local (one process driving several GPUs from one host thread each).
#include "er_common.h"

#include "er_merge_protocol.h"

#include "../../include/er_hip.h"

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only: every function is reached through dlsym

#include <algorithm>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclReduce) Reduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
  char why[256] = "librccl.so.1 not found";      // dlerror() of the failed load, read ONCE (a second call returns NULL)
};

Rccl& rccl_state() {
  static Rccl R;
  return R;
}
const char* rccl_reason() { return rccl_state().why; }

Rccl* rccl() {
  Rccl& R = rccl_state();
  static std::once_flag once;
  std::call_once(once, [&R] {
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : names) {
      R.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (R.lib) break;
      const char* e = dlerror();
      if (e) snprintf(R.why, sizeof R.why, "%s", e);
    }
    if (!R.lib) return;
#define ER_SYM(field, name) R.field = reinterpret_cast<decltype(R.field)>(dlsym(R.lib, name))
    ER_SYM(GetUniqueId, "ncclGetUniqueId");
    ER_SYM(CommInitRank, "ncclCommInitRank");
    ER_SYM(CommInitAll, "ncclCommInitAll");
    ER_SYM(CommDestroy, "ncclCommDestroy");
    ER_SYM(AllReduce, "ncclAllReduce");
    ER_SYM(Reduce, "ncclReduce");
    ER_SYM(AllGather, "ncclAllGather");
    ER_SYM(Send, "ncclSend");
    ER_SYM(Recv, "ncclRecv");
    ER_SYM(GroupStart, "ncclGroupStart");
    ER_SYM(GroupEnd, "ncclGroupEnd");
    ER_SYM(GetErrorString, "ncclGetErrorString");
#undef ER_SYM
    R.ok = R.GetUniqueId && R.CommInitRank && R.CommInitAll && R.CommDestroy && R.AllReduce && R.Reduce && R.AllGather && R.Send && R.Recv &&
           R.GroupStart && R.GroupEnd && R.GetErrorString;
    if (!R.ok) snprintf(R.why, sizeof R.why, "librccl.so.1 lacks one of the twelve nccl* entry points");
  });
  return R.ok ? &R : nullptr;
}


#define ER_NCCL_TRY(R, expr)                                                                        \
  do {                                                                                              \
    ncclResult_t er_r_ = (expr);                                                                    \
    if (er_r_ != ncclSuccess)                                                                       \
      return ::er::fail("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, (R)->GetErrorString(er_r_)); \
  } while (0)

}  // namespace

// ---- loopback communicator (round 5): N ranks = N host threads of ONE process, all on ONE device ----------------------------------------------
// RCCL refuses two ranks on one GPU and the builder's boxes have one, so until round 5 no N > 1 merge had ever touched a device volume.  The loopback
// transport runs the SAME protocol header over the SAME device volumes, export / import kernels and plane buffers; only the wire is replaced: the sum
// reduction is a kernel that adds the ranks' plane buffers in rank order, the point-to-point step is a device-to-device copy, the small host collectives
// go through a mutex / condition-variable barrier.  It is what `bin/Integrate --gpus N --same_device` and tests/test_tsdf_gpu.py use; it says nothing
// about xGMI.
struct ErLoop {
  int world = 1, device = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0, generation = 0;
  std::vector<const int*> iptr;
  std::vector<int> ibuf;
  std::vector<float*> fptr;
  std::vector<const float*> xsend;
  std::vector<size_t> xcount;
  std::vector<std::vector<int>> xto;
  std::vector<std::vector<size_t>> voff, vcnt;                  // exchange_v: every rank's per-receiver offsets and counts
  int pending_vote = 0, result_vote = 0;
  explicit ErLoop(int w, int dev)
      : world(w), device(dev), iptr((size_t)w), fptr((size_t)w), xsend((size_t)w), xcount((size_t)w), xto((size_t)w), voff((size_t)w), vcnt((size_t)w) {}
  // `last` runs inside the critical section of the last arriver.  Returns the OR of every rank's `vote`: how a failure that only ONE rank saw
  // (a kernel launch, a copy) reaches all of them before anybody decides to skip a later step the others would wait in.
  template <class F> int barrier(F last, int vote = 0) {
    std::unique_lock<std::mutex> lk(m);
    const int gen = generation;
    pending_vote |= vote;
    if (++arrived == world) {
      last();
      result_vote = pending_vote;
      pending_vote = 0;
      arrived = 0;
      generation++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
    return result_vote;                                         // (stable until every rank has entered the NEXT barrier, which needs this one to have returned)
  }
};

struct er_comm_s {
  std::shared_ptr<ErLoop> loop;   // set: a loopback communicator (comm stays NULL)
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  float* buf = nullptr;        // [multi-toucher units][sdf*w | w][64^3] planes of the reduction (grow-only)
  size_t buf_units = 0;
  float *sbuf = nullptr, *rbuf = nullptr;   // raw single-toucher units on their way out / in (grow-only)
  size_t sbuf_units = 0, rbuf_units = 0;
  er::MergeStats last;         // what the last merge moved (ring protocol)
  er::OwnerMergeStats last_owner;   // ... (owner protocol)
  int last_impl = 0;           // 0 = ring (er::merge_protocol), 1 = owner (er::merge_protocol_owner)
  float* vbuf[2] = {nullptr, nullptr};      // owner merge: incoming records of step 1 / step 3 (grow-only)
  size_t vbuf_floats[2] = {0, 0};
  float* xbuf = nullptr;       // owner merge: outgoing records (grow-only)
  size_t xbuf_floats = 0;
  int* ikeys = nullptr;        // device scratch: [1 + max_keys * (world + 1)] ints (grow-only)
  size_t ikeys_cap = 0;
};


namespace {

// er::MergeTransport over one RCCL communicator; small host arrays are staged through the communicator's device scratch.
struct RcclTransport : er::MergeTransport {
  Rccl* R;
  er_comm_t c;
  hipStream_t S;
  RcclTransport(Rccl* r, er_comm_t comm, hipStream_t s) : R(r), c(comm), S(s) {}
  int rank() const override { return c->rank; }
  int world() const override { return c->world; }
  int scratch(size_t ints) {
    if (c->ikeys_cap >= ints) return 0;
    if (c->ikeys) (void)hipFree(c->ikeys);
    c->ikeys = nullptr;
    c->ikeys_cap = 0;
    const size_t cap = std::max<size_t>(ints, 64);
    ER_HIP_TRY(hipMalloc((void**)&c->ikeys, cap * sizeof(int)));
    c->ikeys_cap = cap;
    return 0;
  }
  int allreduce_max(int* v, int n) override {
    if (scratch(2 * (size_t)n)) return 1;
    ER_HIP_TRY(hipMemcpyAsync(c->ikeys, v, (size_t)n * sizeof(int), hipMemcpyHostToDevice, S));
    ER_NCCL_TRY(R, R->AllReduce(c->ikeys, c->ikeys + n, (size_t)n, ncclInt32, ncclMax, c->comm, S));
    ER_HIP_TRY(hipMemcpyAsync(v, c->ikeys + n, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, S));
    ER_HIP_TRY(hipStreamSynchronize(S));
    return 0;
  }
  int allgather(const int* mine, int n, int* all) override {
    if (scratch((size_t)n * ((size_t)c->world + 1))) return 1;
    ER_HIP_TRY(hipMemcpyAsync(c->ikeys, mine, (size_t)n * sizeof(int), hipMemcpyHostToDevice, S));
    ER_NCCL_TRY(R, R->AllGather(c->ikeys, c->ikeys + n, (size_t)n, ncclInt32, c->comm, S));
    ER_HIP_TRY(hipMemcpyAsync(all, c->ikeys + n, (size_t)n * c->world * sizeof(int), hipMemcpyDeviceToHost, S));
    ER_HIP_TRY(hipStreamSynchronize(S));
    return 0;
  }
  int reduce_sum(float* planes, size_t count, int root) override {
    if (root < 0)
      ER_NCCL_TRY(R, R->AllReduce(planes, planes, count, ncclFloat32, ncclSum, c->comm, S));   // the only data-path collective
    else
      ER_NCCL_TRY(R, R->Reduce(planes, planes, count, ncclFloat32, ncclSum, root, c->comm, S));
    return 0;
  }
  // the single-toucher units: every send and receive of this rank in ONE group (RCCL pairs them up across the ranks; over xGMI each pair is a
  // direct link).  A rank with nothing to send or receive issues nothing.
  int exchange(const float* send, size_t send_count, const std::vector<int>& send_to, float* recv, const std::vector<size_t>& recv_count) override {
    bool any = !send_to.empty() && send_count > 0;
    for (size_t n : recv_count) any = any || n > 0;
    if (!any) return 0;
    ER_NCCL_TRY(R, R->GroupStart());
    ncclResult_t bad = ncclSuccess;
    if (send_count > 0)
      for (int q : send_to) {
        const ncclResult_t e = R->Send(send, send_count, ncclFloat32, q, c->comm, S);
        if (e != ncclSuccess) bad = e;
      }
    size_t off = 0;
    for (int q = 0; q < c->world; q++) {
      const size_t n = recv_count[(size_t)q];
      if (!n) continue;
      const ncclResult_t e = R->Recv(recv + off, n, ncclFloat32, q, c->comm, S);
      if (e != ncclSuccess) bad = e;
      off += n;
    }
    const ncclResult_t e2 = R->GroupEnd();                      // (always closed, whatever a call inside said)
    if (bad != ncclSuccess || e2 != ncclSuccess)
      return ::er::fail("er_tsdf_allreduce: ncclSend / ncclRecv failed: %s", R->GetErrorString(bad != ncclSuccess ? bad : e2));
    return 0;
  }
  // owner merge: a block of its own per receiver, every send and receive of this rank in ONE group -- over xGMI every (sender, owner) pair is a
  // direct link, so the seven links of a GPU carry their records at the same time (a ring reduction crosses them one after the other)
  int exchange_v(const float* send, const std::vector<size_t>& send_off, const std::vector<size_t>& send_count, float* recv,
                 const std::vector<size_t>& recv_count) override {
    bool any = false;
    for (size_t n : send_count) any = any || n > 0;
    for (size_t n : recv_count) any = any || n > 0;
    if (!any) return 0;
    ER_NCCL_TRY(R, R->GroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int q = 0; q < c->world; q++) {
      if (!send_count[(size_t)q]) continue;
      const ncclResult_t e = R->Send(send + send_off[(size_t)q], send_count[(size_t)q], ncclFloat32, q, c->comm, S);
      if (e != ncclSuccess) bad = e;
    }
    size_t off = 0;
    for (int q = 0; q < c->world; q++) {
      const size_t n = recv_count[(size_t)q];
      if (!n) continue;
      const ncclResult_t e = R->Recv(recv + off, n, ncclFloat32, q, c->comm, S);
      if (e != ncclSuccess) bad = e;
      off += n;
    }
    const ncclResult_t e2 = R->GroupEnd();
    if (bad != ncclSuccess || e2 != ncclSuccess)
      return ::er::fail("er_tsdf_allreduce: ncclSend / ncclRecv failed: %s", R->GetErrorString(bad != ncclSuccess ? bad : e2));
    ER_HIP_TRY(hipStreamSynchronize(S));                        // the owner's next step reads the records from the host's point of view
    return 0;
  }
};

// (RcclTransport::exchange_v is defined below the class: it shares the group logic)
// dst[i] = ((src_0[i] + src_1[i]) + ...) + src_{n-1}[i]: the ranks' plane buffers added in rank order (dst may be one of the sources)
struct LoopSrc { const float* p[16]; };
__global__ __launch_bounds__(256) void k_loop_sum(LoopSrc S, int n, float* __restrict__ dst, size_t count) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    float a = S.p[0][i];
    for (int q = 1; q < n; q++) a += S.p[q][i];
    dst[i] = a;
  }
}

// er::MergeTransport over an ErLoop: every rank is a host thread of this process, every buffer lives on the one device.
struct LoopTransport : er::MergeTransport {
  er_comm_t c;
  ErLoop& L;
  hipStream_t S;
  LoopTransport(er_comm_t comm, hipStream_t s) : c(comm), L(*comm->loop), S(s) {}
  int rank() const override { return c->rank; }
  int world() const override { return c->world; }
  int allreduce_max(int* v, int n) override {
    L.iptr[(size_t)c->rank] = v;
    L.barrier([&] {
      L.ibuf.assign((size_t)n, -2147483647 - 1);
      for (int q = 0; q < L.world; q++)
        for (int i = 0; i < n; i++) L.ibuf[(size_t)i] = std::max(L.ibuf[(size_t)i], L.iptr[(size_t)q][i]);
    });
    for (int i = 0; i < n; i++) v[i] = L.ibuf[(size_t)i];
    L.barrier([] {});                                           // nobody overwrites ibuf before everyone has read it
    return 0;
  }
  int allgather(const int* mine, int n, int* all) override {
    L.iptr[(size_t)c->rank] = mine;
    L.barrier([&] {
      L.ibuf.resize((size_t)n * L.world);
      for (int q = 0; q < L.world; q++) std::copy(L.iptr[(size_t)q], L.iptr[(size_t)q] + n, L.ibuf.begin() + (size_t)q * n);
    });
    std::copy(L.ibuf.begin(), L.ibuf.end(), all);
    L.barrier([] {});
    return 0;
  }
  int reduce_sum(float* planes, size_t count, int root) override {
    if (L.world > 16) return ::er::fail("er_tsdf_allreduce: the loopback communicator holds at most 16 ranks");
    int rc = 0;
    if (hipStreamSynchronize(S) != hipSuccess) rc = 1;         // this rank's export kernel has written `planes`
    L.fptr[(size_t)c->rank] = planes;
    rc = L.barrier([] {}, rc);                                  // (from here on rc is the SAME on every rank: nobody skips a barrier the others wait in)
    const int owner = root < 0 ? 0 : root;                      // the rank that adds; with root < 0 the others copy its result
    int mine = 0;
    if (c->rank == owner && !rc) {
      LoopSrc src;
      for (int q = 0; q < 16; q++) src.p[q] = q < L.world ? L.fptr[(size_t)q] : nullptr;
      const unsigned blocks = (unsigned)std::min<size_t>((count + 255) / 256, 16384);
      hipLaunchKernelGGL(k_loop_sum, dim3(blocks), dim3(256), 0, S, src, L.world, planes, count);
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(S) != hipSuccess) mine = 1;
    }
    rc = L.barrier([] {}, rc | mine);
    mine = 0;
    if (root < 0 && c->rank != owner && !rc) {
      if (hipMemcpyAsync(planes, L.fptr[(size_t)owner], count * sizeof(float), hipMemcpyDeviceToDevice, S) != hipSuccess || hipStreamSynchronize(S) != hipSuccess) mine = 1;
    }
    rc = L.barrier([] {}, rc | mine);                           // the owner's buffer stays untouched until everybody has copied
    return rc ? ::er::fail("er_tsdf_allreduce (loopback): a rank of the communicator failed in the sum step (%s)", hipGetErrorString(hipGetLastError())) : 0;
  }
  int exchange(const float* send, size_t send_count, const std::vector<int>& send_to, float* recv, const std::vector<size_t>& recv_count) override {
    int rc = 0;
    if (hipStreamSynchronize(S) != hipSuccess) rc = 1;         // this rank's raw export has written `send`
    L.xsend[(size_t)c->rank] = send;
    L.xcount[(size_t)c->rank] = send_count;
    L.xto[(size_t)c->rank] = send_to;
    rc = L.barrier([] {}, rc);
    int mine = 0;
    size_t off = 0;
    for (int q = 0; q < L.world && !rc && !mine; q++) {
      const size_t n = recv_count[(size_t)q];
      if (!n) continue;
      const std::vector<int>& to = L.xto[(size_t)q];           // what I expect from q must be what q sends to me
      if (q == c->rank || n != L.xcount[(size_t)q] || std::find(to.begin(), to.end(), c->rank) == to.end()) mine = 2;
      else if (hipMemcpyAsync(recv + off, L.xsend[(size_t)q], n * sizeof(float), hipMemcpyDeviceToDevice, S) != hipSuccess) mine = 1;
      off += n;
    }
    if (!rc && !mine && hipStreamSynchronize(S) != hipSuccess) mine = 1;
    rc = L.barrier([] {}, rc | mine);                           // the senders' blocks stay untouched until everybody has copied
    if (rc & 2) return ::er::fail("er_tsdf_allreduce (loopback): the ranks disagree about who sends what");
    return rc ? ::er::fail("er_tsdf_allreduce (loopback): a rank of the communicator failed in the point-to-point step (%s)", hipGetErrorString(hipGetLastError())) : 0;
  }
  int exchange_v(const float* send, const std::vector<size_t>& send_off, const std::vector<size_t>& send_count, float* recv,
                 const std::vector<size_t>& recv_count) override {
    int rc = 0;
    if (hipStreamSynchronize(S) != hipSuccess) rc = 1;         // this rank's records are written
    L.xsend[(size_t)c->rank] = send;
    L.voff[(size_t)c->rank] = send_off;
    L.vcnt[(size_t)c->rank] = send_count;
    rc = L.barrier([] {}, rc);
    int mine = 0;
    size_t off = 0;
    for (int q = 0; q < L.world && !rc && !mine; q++) {
      const size_t n = recv_count[(size_t)q];
      if (!n) continue;
      if (q == c->rank || n != L.vcnt[(size_t)q][(size_t)c->rank]) mine = 2;   // what I expect from q must be what q sends to me
      else if (hipMemcpyAsync(recv + off, L.xsend[(size_t)q] + L.voff[(size_t)q][(size_t)c->rank], n * sizeof(float), hipMemcpyDeviceToDevice, S) != hipSuccess) mine = 1;
      off += n;
    }
    for (int q = 0; q < L.world && !rc && !mine; q++)          // ... and nobody may send me what I do not expect
      if (q != c->rank && L.vcnt[(size_t)q][(size_t)c->rank] != recv_count[(size_t)q]) mine = 2;
    if (!rc && !mine && hipStreamSynchronize(S) != hipSuccess) mine = 1;
    rc = L.barrier([] {}, rc | mine);
    if (rc & 2) return ::er::fail("er_tsdf_allreduce (loopback): the ranks disagree about who sends what");
    return rc ? ::er::fail("er_tsdf_allreduce (loopback): a rank of the communicator failed in the record exchange (%s)", hipGetErrorString(hipGetLastError())) : 0;
  }
};

// er::MergeVolume over an er_tsdf_t; the planes live in the communicator's grow-only device buffer.
struct DeviceVolume : er::OwnerMergeVolume {
  er_tsdf_t h;
  er_comm_t c;
  DeviceVolume(er_tsdf_t vol, er_comm_t comm) : h(vol), c(comm) {}
  size_t unit_voxels() const override { return ER_UNIT_VOX; }
  int touched_keys(std::vector<int>& keys) override {
    if (er_tsdf_synchronize(h)) return 1;                       // the volume's own streams are drained once per job
    int n = 0;
    if (er_tsdf_unit_count(h, &n)) return 1;                    // fails when the unit pool / hash table overflowed
    keys.assign((size_t)std::max(n, 1), -1);
    if (n > 0 && er_tsdf_unit_keys(h, keys.data())) return 1;
    keys.resize((size_t)n);
    return 0;
  }
  int export_planes(const int* uk, int nu, float** planes) override {
    if (c->buf_units < (size_t)nu) {
      if (c->buf) (void)hipFree(c->buf);
      c->buf = nullptr;
      c->buf_units = 0;
      ER_HIP_TRY(hipMalloc((void**)&c->buf, (size_t)nu * 2 * ER_UNIT_VOX * sizeof(float)));
      c->buf_units = (size_t)nu;
    }
    *planes = c->buf;
    return er_tsdf_export_weighted(h, uk, nu, c->buf);          // on the volume's stream, like the collectives
  }
  int import_planes(const int* uk, int nu, const float* planes) override { return er_tsdf_import_weighted(h, uk, nu, planes); }
  static int grow(float** b, size_t* have, int units) {
    if (*have >= (size_t)units) return 0;
    if (*b) (void)hipFree(*b);
    *b = nullptr;
    *have = 0;
    ER_HIP_TRY(hipMalloc((void**)b, (size_t)units * 2 * ER_UNIT_VOX * sizeof(float)));
    *have = (size_t)units;
    return 0;
  }
  int export_raw(const int* uk, int nu, float** block) override {
    if (grow(&c->sbuf, &c->sbuf_units, nu)) return 1;
    *block = c->sbuf;
    return er_tsdf_export_raw(h, uk, nu, c->sbuf);
  }
  int receive_buffer(int nu, float** block) override {
    if (grow(&c->rbuf, &c->rbuf_units, nu)) return 1;
    *block = c->rbuf;
    return 0;
  }
  int import_raw(const int* uk, int nu, const float* block) override { return er_tsdf_import_raw(h, uk, nu, block); }
  // ---- owner merge: band records (er_tsdf.hip) ----
  static int grow_floats(float** b, size_t* have, size_t floats) {
    if (*have >= floats) return 0;
    if (*b) (void)hipFree(*b);
    *b = nullptr;
    *have = 0;
    const size_t cap = floats + floats / 8 + 1024;
    ER_HIP_TRY(hipMalloc((void**)b, cap * sizeof(float)));
    *have = cap;
    return 0;
  }
  // (the protocol's "count" of a unit is its record size in words here: the owner is the toucher with the LARGEST record, the plan needs sizes only)
  int band_counts(const int* uk, int nu, int* counts) override { return er_tsdf_band_sizes(h, uk, nu, counts); }
  size_t band_record_floats(int count) const override { return (size_t)count; }
  int export_band(const int* uk, const int* counts, int nu, float** block) override {
    size_t total = 0;
    for (int i = 0; i < nu; i++) total += band_record_floats(counts[i]);
    if (grow_floats(&c->xbuf, &c->xbuf_floats, total)) return 1;
    *block = c->xbuf;
    return er_tsdf_export_band(h, uk, counts, nu, c->xbuf);
  }
  int band_receive_buffer(size_t floats, int which, float** block) override {
    if (grow_floats(&c->vbuf[which & 1], &c->vbuf_floats[which & 1], floats)) return 1;
    *block = c->vbuf[which & 1];
    return 0;
  }
  int merge_band(const int* uk, int nu, const std::vector<std::vector<const float*>>& src, const int* self_pos) override {
    std::vector<int> nsrc((size_t)nu);
    std::vector<const void*> recs((size_t)nu * 16, nullptr);
    for (int i = 0; i < nu; i++) {
      nsrc[(size_t)i] = (int)src[(size_t)i].size();
      if (nsrc[(size_t)i] > 16) return ::er::fail("er_tsdf_allreduce: unit %d has %d touchers besides its owner (at most 16)", uk[i], nsrc[(size_t)i]);
      for (int k = 0; k < nsrc[(size_t)i]; k++) recs[(size_t)i * 16 + (size_t)k] = src[(size_t)i][(size_t)k];
    }
    return er_tsdf_merge_band(h, uk, nu, nsrc.data(), self_pos, recs.data());
  }
  int import_band(const int* uk, int nu, const std::vector<const float*>& recs) override {
    std::vector<const void*> r(recs.begin(), recs.end());
    return er_tsdf_import_band(h, uk, nu, r.data());
  }
  int drop_units(const int* uk, int nu) override { return er_tsdf_drop_units(h, uk, nu); }
};

}  // namespace

// The int scratch every small collective of the merge stages through, allocated when the communicator is made: 64 Ki ints hold the
// padded keys of a full 512-unit region from 100+ ranks, so er_tsdf_allreduce itself never allocates before its first collective.
static int comm_scratch(er_comm_t c) {
  ER_HIP_TRY(hipSetDevice(c->device));
  ER_HIP_TRY(hipMalloc((void**)&c->ikeys, (size_t)65536 * sizeof(int)));
  c->ikeys_cap = 65536;
  return 0;
}

static_assert(ER_COMM_ID_BYTES == sizeof(ncclUniqueId), "ER_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");

extern "C" {

int er_comm_unique_id(unsigned char id[ER_COMM_ID_BYTES]) {
  if (!id) return er::fail("er_comm_unique_id: NULL argument");
  Rccl* R = rccl();
  if (!R) return er::fail("er_comm_unique_id: librccl.so.1 could not be loaded (%s)", rccl_reason());
  ncclUniqueId u;
  ER_NCCL_TRY(R, R->GetUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return 0;
}

int er_comm_create(const unsigned char id[ER_COMM_ID_BYTES], int rank, int world, int device, er_comm_t* out) {
  if (!id || !out) return er::fail("er_comm_create: NULL argument");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world) return er::fail("er_comm_create: rank %d not in [0,%d)", rank, world);
  Rccl* R = rccl();
  if (!R) return er::fail("er_comm_create: librccl.so.1 could not be loaded (%s)", rccl_reason());
  ER_HIP_TRY(hipSetDevice(device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  er_comm_t c = new er_comm_s();
  c->rank = rank;
  c->world = world;
  c->device = device;
  ncclResult_t r = R->CommInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) {
    delete c;
    return er::fail("er_comm_create: ncclCommInitRank failed: %s", R->GetErrorString(r));
  }
  if (comm_scratch(c)) {
    (void)R->CommDestroy(c->comm);
    delete c;
    return 1;
  }
  *out = c;
  return 0;
}

int er_comm_create_local(int n, const int* devices, er_comm_t* out) {
  if (n < 1 || !devices || !out) return er::fail("er_comm_create_local: bad arguments");
  for (int i = 0; i < n; i++) out[i] = nullptr;
  Rccl* R = rccl();
  if (!R) return er::fail("er_comm_create_local: librccl.so.1 could not be loaded (%s)", rccl_reason());
  std::vector<ncclComm_t> comms((size_t)n, nullptr);
  ER_NCCL_TRY(R, R->CommInitAll(comms.data(), n, devices));
  for (int i = 0; i < n; i++) {
    er_comm_t c = new er_comm_s();
    c->comm = comms[(size_t)i];
    c->rank = i;
    c->world = n;
    c->device = devices[i];
    out[i] = c;
  }
  for (int i = 0; i < n; i++)
    if (comm_scratch(out[i])) {                                  // all or nothing, like every other failure path of this call (the
      const std::string why = er_last_error();                   // handles used to stay in out[] for the caller to destroy, the earlier paths left NULLs)
      for (int k = 0; k < n; k++) {
        er_comm_destroy(out[k]);
        out[k] = nullptr;
      }
      return er::fail("%s", why.c_str());
    }
  return 0;
}

int er_comm_create_loopback(int n, int device, er_comm_t* out) {
  if (n < 1 || n > 16 || !out) return er::fail("er_comm_create_loopback: 1 .. 16 ranks and a handle array are needed");
  for (int i = 0; i < n; i++) out[i] = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return er::fail("er_comm_create_loopback: device %d is not there", device);
  std::shared_ptr<ErLoop> L = std::make_shared<ErLoop>(n, device);
  for (int i = 0; i < n; i++) {
    er_comm_t c = new er_comm_s();
    c->loop = L;
    c->rank = i;
    c->world = n;
    c->device = device;
    out[i] = c;
  }
  return 0;
}

int er_comm_destroy(er_comm_t c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  if (c->buf) (void)hipFree(c->buf);
  if (c->sbuf) (void)hipFree(c->sbuf);
  if (c->rbuf) (void)hipFree(c->rbuf);
  if (c->xbuf) (void)hipFree(c->xbuf);
  if (c->vbuf[0]) (void)hipFree(c->vbuf[0]);
  if (c->vbuf[1]) (void)hipFree(c->vbuf[1]);
  if (c->ikeys) (void)hipFree(c->ikeys);
  if (c->comm) {
    Rccl* R = rccl();
    if (R) (void)R->CommDestroy(c->comm);
  }
  delete c;
  return 0;
}

int er_comm_rank(er_comm_t c) { return c ? c->rank : -1; }
int er_comm_world(er_comm_t c) { return c ? c->world : -1; }

// The frame-split merge (see the file header): er_merge_protocol.h's steps over RCCL and the device-resident volume.  Every
// rank of the communicator calls it once, each from its own host thread / process; root < 0 leaves the merged volume on
// every rank, otherwise only on `root`.  A rank-local failure -- a bad `root`, a volume that lives on another device than the
// communicator, a unit pool / hash table overflow, a failed allocation of the plane buffer -- is carried through the next collective
// as a status, so ALL ranks return nonzero together instead of the healthy ones waiting for ever (the small device scratch the
// status travels in is allocated by er_comm_create*, not here).  What stays fatal for the whole communicator: a NULL handle, RCCL
// missing, and a HIP / RCCL error inside a collective itself (MERGE_TRANSPORT_FAILURE) -- after those the peers' state is unknown.
int er_tsdf_allreduce(er_tsdf_t h, er_comm_t c, int root, int* union_units) {
  if (!h || !c) return er::fail("er_tsdf_allreduce: NULL argument");
  Rccl* R = c->loop ? nullptr : rccl();
  if (!c->loop && !R) return er::fail("er_tsdf_allreduce: librccl.so.1 could not be loaded (%s)", rccl_reason());
  // which protocol: the owner merge (round 6; reduce-scatter by unit, band records, rank-ordered sums) unless ER_MERGE_IMPL=ring asks for round 5's
  // ring reduction of whole planes; a distributed result (root == ER_MERGE_DISTRIBUTED) only exists in the owner merge
  const char* impl_env = getenv("ER_MERGE_IMPL");
  const bool ring = impl_env && std::string(impl_env) == "ring";
  int pre = 0;
  if (root >= c->world || root < ER_MERGE_DISTRIBUTED) pre = er::fail("er_tsdf_allreduce: root %d not in [0,%d) and neither ER_MERGE_ALL nor ER_MERGE_DISTRIBUTED", root, c->world);
  else if (ring && root == ER_MERGE_DISTRIBUTED) pre = er::fail("er_tsdf_allreduce: ER_MERGE_IMPL=ring cannot leave the result distributed (root = ER_MERGE_DISTRIBUTED)");
  else if (!ring && c->world > 17) pre = er::fail("er_tsdf_allreduce: the owner merge adds at most 17 ranks per unit (ER_MERGE_IMPL=ring has no such limit)");
  else if (er::tsdf_device(h) != c->device)
    pre = er::fail("er_tsdf_allreduce: the volume lives on device %d, the communicator on %d", er::tsdf_device(h), c->device);
  const hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess && !pre) pre = er::fail("er_tsdf_allreduce: hipSetDevice(%d): %s", c->device, hipGetErrorString(e));
  std::string why = pre ? er_last_error() : "";                 // (the protocol's later steps may overwrite the thread's message)
  hipStream_t stream = pre && er::tsdf_device(h) != c->device ? nullptr : er::tsdf_stream(h);
  RcclTransport t_rccl(R, c, stream);
  std::unique_ptr<LoopTransport> t_loop(c->loop ? new LoopTransport(c, stream) : nullptr);
  er::MergeTransport& t = c->loop ? static_cast<er::MergeTransport&>(*t_loop) : static_cast<er::MergeTransport&>(t_rccl);
  DeviceVolume v(h, c);
  c->last = er::MergeStats();
  c->last_owner = er::OwnerMergeStats();
  c->last_impl = ring ? 0 : 1;
  const int r = ring ? er::merge_protocol(t, v, root, union_units, pre ? 1 : 0, &c->last)
                     : er::merge_protocol_owner(t, v, root, union_units, pre ? 1 : 0, &c->last_owner);
  if (r == er::MERGE_OK) {
    ER_HIP_TRY(hipStreamSynchronize(er::tsdf_stream(h)));
    return 0;
  }
  if (r == er::MERGE_PEER_FAILURE)
    return er::fail("er_tsdf_allreduce: another rank of the communicator failed before the merge (its own er_last_error() says why); nothing was merged");
  if (pre) return er::fail("%s", why.c_str());
  return 1;                                                     // local / transport failure: the message is already recorded
}

int er_comm_merge_stats_owner(er_comm_t c, long long stats[12]) {
  if (!c || !stats) return er::fail("er_comm_merge_stats_owner: NULL argument");
  const er::OwnerMergeStats& m = c->last_owner;
  stats[0] = c->last_impl;
  stats[1] = m.union_units; stats[2] = m.multi_units; stats[3] = m.single_units; stats[4] = m.owned_units; stats[5] = m.owned_multi; stats[6] = m.dropped_units;
  stats[7] = (long long)(m.sent_floats[0] * sizeof(float));
  stats[8] = (long long)(m.received_floats[0] * sizeof(float));
  stats[9] = (long long)(m.sent_floats[1] * sizeof(float));
  stats[10] = (long long)(m.received_floats[1] * sizeof(float));
  stats[11] = (long long)(m.dense_floats * sizeof(float));
  return 0;
}

int er_comm_merge_stats(er_comm_t c, long long stats[8]) {
  if (!c || !stats) return er::fail("er_comm_merge_stats: NULL argument");
  if (c->last_impl == 1) {                                        // the owner merge in the ring protocol's terms
    const er::OwnerMergeStats& o = c->last_owner;
    stats[0] = o.union_units; stats[1] = o.multi_units; stats[2] = o.single_units; stats[3] = o.dropped_units; stats[4] = o.owned_multi;
    stats[5] = 0;                                                 // nothing goes through a reduction collective
    stats[6] = (long long)((o.sent_floats[0] + o.sent_floats[1]) * sizeof(float));
    stats[7] = (long long)((o.received_floats[0] + o.received_floats[1]) * sizeof(float));
    return 0;
  }
  const er::MergeStats& m = c->last;
  stats[0] = m.union_units; stats[1] = m.multi_units; stats[2] = m.single_units; stats[3] = m.sent_units; stats[4] = m.received_units;
  stats[5] = (long long)(m.reduced_floats * sizeof(float));
  stats[6] = (long long)(m.sent_floats * sizeof(float));
  stats[7] = (long long)(m.received_floats * sizeof(float));
  return 0;
}

void er_frame_block(int n_frames, int rank, int world, int* lo, int* hi) {
  const int per = world > 0 ? (n_frames + world - 1) / world : n_frames;
  const int a = std::min(rank * per, n_frames);
  if (lo) *lo = a;
  if (hi) *hi = std::min(a + per, n_frames);
}

}  // extern "C"
