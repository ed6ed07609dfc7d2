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
//   3. fixed-size all-gather of the keys padded with -1 to that count -> sorted union AND, since round 5, who touched what:
//      the union splits into MULTI-toucher units (frames of two or more ranks met there: TSDFVolume.cpp:93-94 is a sum only for
//      these) and SINGLE-toucher units (one rank holds the final voxels already);
//   4. local: export the [key][sdf*weight | weight] planes of the multi-toucher units (a rank that never touched one of them
//      contributes zeros) and the RAW [key][sdf | weight] planes of the rank's own single-toucher units that have to travel;
//      all-reduce(MAX) of that step's status, same reason as in 2;
//   5. ONE sum reduction over the multi-toucher planes -- to `root`, or to everybody -- the only arithmetic collective of the
//      pipeline, and ONE point-to-point exchange that carries every single-toucher unit, untouched, from its owner to `root` (or,
//      root < 0, to every other rank); a unit that already sits where it is wanted does not move at all;
//   6. import on the receiving rank(s): multi-toucher units weight = W, sdf = SW / W; single-toucher units as they were.
// Steps 2-4 move a few hundred ints; step 5 moves 2 MiB per multi-toucher unit through the reduction and 2 MiB per travelling
// single-toucher unit once over one link.  Rounds 2-4 reduced the planes of the WHOLE union (2.3 GB for the 1103 units of configs[3],
// zeros for every unit a rank never saw): on a drifting path most units belong to one contiguous frame block, and a single-toucher
// unit that went through sdf * w / w came back rounded where it now arrives bit for bit.
#pragma once

#include <algorithm>
#include <cstddef>
#include <utility>
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
  // Point-to-point step: this rank sends the SAME block send[0 .. send_count) to every rank listed in send_to and receives recv_count[q] floats from
  // rank q (0 = nothing) into recv, the blocks in rank order.  Called by every rank (possibly with nothing to send and nothing to receive).
  virtual int exchange(const float* send, size_t send_count, const std::vector<int>& send_to, float* recv, const std::vector<size_t>& recv_count) = 0;
};

// What the protocol needs from one rank's volume.
struct MergeVolume {
  virtual ~MergeVolume() {}
  virtual int touched_keys(std::vector<int>& keys) = 0;                                   // nonzero = local failure (message already recorded)
  virtual int export_planes(const int* keys, int n, float** planes) = 0;                 // [n][sdf*w | w][unit voxels], where the transport reduces
  virtual int import_planes(const int* keys, int n, const float* planes) = 0;
  // raw units: [n][sdf | w][unit voxels], bit for bit.  export_raw fills a send block, receive_buffer provides room for n incoming units (both in the
  // memory space the transport moves), import_raw creates / overwrites the units.
  virtual int export_raw(const int* keys, int n, float** block) = 0;
  virtual int receive_buffer(int n, float** block) = 0;
  virtual int import_raw(const int* keys, int n, const float* block) = 0;
  virtual size_t unit_voxels() const = 0;
};

enum { MERGE_OK = 0, MERGE_LOCAL_FAILURE = 1, MERGE_PEER_FAILURE = 2, MERGE_TRANSPORT_FAILURE = 3 };

// What one merge moved, as seen by this rank (floats = 4 bytes each; a unit = 2 * unit_voxels floats).
struct MergeStats {
  int union_units = 0;          // size of the key union
  int multi_units = 0;          // units two or more ranks touched: went through the sum reduction
  int single_units = 0;         // units exactly one rank touched
  int sent_units = 0;           // this rank's single-toucher units that travelled (counted once, whatever the number of receivers)
  int received_units = 0;       // single-toucher units of other ranks that arrived here
  size_t reduced_floats = 0;    // count handed to reduce_sum
  size_t sent_floats = 0, received_floats = 0;
};

// Returns MERGE_OK, or -- on EVERY rank, after the same collective -- which kind of failure stopped the merge.
inline int merge_protocol(MergeTransport& t, MergeVolume& v, int root, int* union_units, int pre_status = 0, MergeStats* stats = nullptr) {
  if (union_units) *union_units = 0;
  MergeStats st;
  std::vector<int> keys;
  const int st1 = (pre_status || v.touched_keys(keys)) ? 1 : 0;
  if (st1) keys.clear();
  int agree[2] = {(int)keys.size(), st1};
  if (t.allreduce_max(agree, 2)) return MERGE_TRANSPORT_FAILURE;
  if (agree[1]) return st1 ? MERGE_LOCAL_FAILURE : MERGE_PEER_FAILURE;
  const int max_keys = agree[0];
  if (max_keys <= 0) {                                                 // nobody touched anything
    if (stats) *stats = st;
    return MERGE_OK;
  }
  const int W = t.world(), me = t.rank();
  std::vector<int> padded((size_t)max_keys, -1), all((size_t)max_keys * (size_t)W, -1);
  std::copy(keys.begin(), keys.end(), padded.begin());
  if (t.allgather(padded.data(), max_keys, all.data())) return MERGE_TRANSPORT_FAILURE;
  // (key, rank) pairs of everything anybody touched, sorted by key: a key's run length is its toucher count (identical on every rank)
  std::vector<std::pair<int, int>> kr;
  for (int q = 0; q < W; q++) {
    std::vector<int> seg(all.begin() + (size_t)q * max_keys, all.begin() + (size_t)(q + 1) * max_keys);
    std::sort(seg.begin(), seg.end());
    seg.erase(std::unique(seg.begin(), seg.end()), seg.end());         // (a volume lists a unit once; be safe)
    for (int k : seg)
      if (k >= 0) kr.push_back(std::make_pair(k, q));
  }
  std::sort(kr.begin(), kr.end());
  std::vector<int> multi, mine_to_send;                                // multi-toucher keys; my single-toucher keys that have to travel
  std::vector<std::vector<int>> from((size_t)W);                       // single-toucher keys that arrive here, by owner
  int nu = 0, travelling = 0;                                          // (travelling: single-toucher units that move at all -- the same number on every rank)
  for (size_t i = 0; i < kr.size();) {
    size_t j = i;
    while (j < kr.size() && kr[j].first == kr[i].first) j++;
    nu++;
    if (j - i >= 2) {
      multi.push_back(kr[i].first);
    } else {
      st.single_units++;
      const int owner = kr[i].second;
      const bool travels = root < 0 ? W > 1 : owner != root;           // to everybody else / to the root unless it lives there
      travelling += travels ? 1 : 0;
      if (travels && owner == me) mine_to_send.push_back(kr[i].first);
      if (travels && owner != me && (root < 0 || root == me)) from[(size_t)owner].push_back(kr[i].first);
    }
    i = j;
  }
  st.union_units = nu;
  st.multi_units = (int)multi.size();
  if (union_units) *union_units = nu;
  if (nu == 0) {
    if (stats) *stats = st;
    return MERGE_OK;
  }
  const size_t unit_floats = 2 * v.unit_voxels();
  std::vector<int> incoming;                                           // in rank order, like the blocks of the exchange
  std::vector<size_t> recv_count((size_t)W, 0);
  for (int q = 0; q < W; q++) {
    incoming.insert(incoming.end(), from[(size_t)q].begin(), from[(size_t)q].end());
    recv_count[(size_t)q] = from[(size_t)q].size() * unit_floats;
  }
  std::vector<int> send_to;
  if (!mine_to_send.empty()) {
    if (root < 0) {
      for (int q = 0; q < W; q++)
        if (q != me) send_to.push_back(q);
    } else {
      send_to.push_back(root);
    }
  }
  float *planes = nullptr, *send = nullptr, *recv = nullptr;
  int st2 = 0;
  if (!multi.empty() && v.export_planes(multi.data(), (int)multi.size(), &planes)) st2 = 1;
  if (!st2 && !mine_to_send.empty() && v.export_raw(mine_to_send.data(), (int)mine_to_send.size(), &send)) st2 = 1;
  if (!st2 && !incoming.empty() && v.receive_buffer((int)incoming.size(), &recv)) st2 = 1;
  int any = st2;
  if (t.allreduce_max(&any, 1)) return MERGE_TRANSPORT_FAILURE;
  if (any) return st2 ? MERGE_LOCAL_FAILURE : MERGE_PEER_FAILURE;
  if (!multi.empty()) {                                                // (the same decision on every rank: `multi` is a function of the gathered keys)
    st.reduced_floats = multi.size() * unit_floats;
    if (t.reduce_sum(planes, st.reduced_floats, root)) return MERGE_TRANSPORT_FAILURE;
  }
  if (travelling > 0) {                                                // (likewise: every rank enters the exchange, or none)
    st.sent_units = (int)mine_to_send.size();
    st.received_units = (int)incoming.size();
    st.sent_floats = mine_to_send.size() * unit_floats;
    st.received_floats = incoming.size() * unit_floats;
    if (t.exchange(send, st.sent_floats, send_to, recv, recv_count)) return MERGE_TRANSPORT_FAILURE;
  }
  if (stats) *stats = st;
  if (root < 0 || root == me) {                                        // after the last collective: nobody waits for this rank
    if (!multi.empty() && v.import_planes(multi.data(), (int)multi.size(), planes)) return MERGE_LOCAL_FAILURE;
    if (!incoming.empty() && v.import_raw(incoming.data(), (int)incoming.size(), recv)) return MERGE_LOCAL_FAILURE;
  }
  return MERGE_OK;
}

}  // namespace er
