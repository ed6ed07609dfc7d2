// er_merge_protocol.h -- the frame-split merge of path A (the algebra of TSDFVolume.cpp:93-94 applied as a sum) as
// plain host C++ over two small interfaces, so that the SAME protocol code runs
//   * on RCCL + a device-resident volume          (er_multi.hip: er_tsdf_allreduce, what bin/Integrate --gpus N and bench.py use)
//   * on host threads + host arrays, world = 2, 3 (tests/cpp/merge_protocol_check.cpp: the CPU test of this very code).
//
// Steps (every rank of the communicator executes them in the same order; root < 0 = the merged volume on every rank):
//   1. local: the keys of the units this rank touched.  A LOCAL failure here (unit pool or hash table overflowed, allocation
//      failed) -- or one the caller found BEFORE the protocol started and hands in as `pre_status` (a bad argument, a volume
//      on another device than its communicator) -- must not make the rank leave: the others would wait in the next
//      collective for ever.  So it becomes a status;
//   2. all-reduce(MAX) of { key count, status }: every rank learns the padded key count and whether ANY rank failed; if one
//      did, all of them return an error together, after that collective;
//   3. fixed-size all-gather of the keys padded with -1 to that count -> sorted union (identical on every rank);
//   4. local: export the [key][sdf*weight | weight] planes of the union (units a rank never touched contribute zeros);
//      all-reduce(MAX) of that step's status, same reason as in 2;
//   5. ONE sum reduction over the planes -- to `root`, or to everybody -- the only data-path collective of the pipeline;
//   6. import on the receiving rank(s): weight = W, sdf = SW / W.
// Steps 2-4 move a few hundred ints; step 5 moves 2 MiB per unit of the union.
#pragma once

#include <algorithm>
#include <cstddef>
#include <vector>

namespace er {

// Collectives over the ranks of one communicator.  Every method is called by every rank, in the same order.
struct MergeTransport {
  virtual ~MergeTransport() {}
  virtual int rank() const = 0;
  virtual int world() const = 0;
  virtual int allreduce_max(int* v, int n) = 0;                       // host ints, in place
  virtual int allgather(const int* mine, int n, int* all) = 0;        // n host ints per rank -> world * n, in rank order
  virtual int reduce_sum(float* planes, size_t count, int root) = 0;  // in place, in the memory space export_planes returns; root < 0: all-reduce
};

// What the protocol needs from one rank's volume.
struct MergeVolume {
  virtual ~MergeVolume() {}
  virtual int touched_keys(std::vector<int>& keys) = 0;                                   // nonzero = local failure (message already recorded)
  virtual int export_planes(const int* union_keys, int nu, float** planes) = 0;          // [nu][2][unit voxels], where the transport reduces
  virtual int import_planes(const int* union_keys, int nu, const float* planes) = 0;
  virtual size_t unit_voxels() const = 0;
};

enum { MERGE_OK = 0, MERGE_LOCAL_FAILURE = 1, MERGE_PEER_FAILURE = 2, MERGE_TRANSPORT_FAILURE = 3 };

// Returns MERGE_OK, or -- on EVERY rank, after the same collective -- which kind of failure stopped the merge.
inline int merge_protocol(MergeTransport& t, MergeVolume& v, int root, int* union_units, int pre_status = 0) {
  if (union_units) *union_units = 0;
  std::vector<int> keys;
  const int st1 = (pre_status || v.touched_keys(keys)) ? 1 : 0;
  if (st1) keys.clear();
  int agree[2] = {(int)keys.size(), st1};
  if (t.allreduce_max(agree, 2)) return MERGE_TRANSPORT_FAILURE;
  if (agree[1]) return st1 ? MERGE_LOCAL_FAILURE : MERGE_PEER_FAILURE;
  const int max_keys = agree[0];
  if (max_keys <= 0) return MERGE_OK;                                  // nobody touched anything
  std::vector<int> padded((size_t)max_keys, -1), all((size_t)max_keys * (size_t)t.world(), -1);
  std::copy(keys.begin(), keys.end(), padded.begin());
  if (t.allgather(padded.data(), max_keys, all.data())) return MERGE_TRANSPORT_FAILURE;
  std::sort(all.begin(), all.end());
  all.erase(std::unique(all.begin(), all.end()), all.end());
  all.erase(std::remove_if(all.begin(), all.end(), [](int k) { return k < 0; }), all.end());
  const int nu = (int)all.size();
  if (union_units) *union_units = nu;
  if (nu == 0) return MERGE_OK;
  float* planes = nullptr;
  int st2 = v.export_planes(all.data(), nu, &planes) ? 1 : 0;
  int any = st2;
  if (t.allreduce_max(&any, 1)) return MERGE_TRANSPORT_FAILURE;
  if (any) return st2 ? MERGE_LOCAL_FAILURE : MERGE_PEER_FAILURE;
  if (t.reduce_sum(planes, (size_t)nu * 2 * v.unit_voxels(), root)) return MERGE_TRANSPORT_FAILURE;
  if (root < 0 || root == t.rank())
    if (v.import_planes(all.data(), nu, planes)) return MERGE_LOCAL_FAILURE;   // after the last collective: nobody waits for this rank
  return MERGE_OK;
}

}  // namespace er
