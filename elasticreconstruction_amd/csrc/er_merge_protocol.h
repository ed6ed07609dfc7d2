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
  // The same step with a block of its own per receiver (owner merge, round 6): this rank sends send[send_off[q] .. + send_count[q]) to rank q
  // (count 0 = nothing; blocks may coincide) and receives recv_count[q] floats from rank q into recv, the blocks in rank order.  Every rank calls
  // it; a failure on ONE rank comes back nonzero on EVERY rank (the step ends with an agreement on its status).
  virtual int exchange_v(const float* send, const std::vector<size_t>& send_off, const std::vector<size_t>& send_count, float* recv,
                         const std::vector<size_t>& recv_count) = 0;
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


// ---- round 6: the OWNER merge (reduce-scatter by unit, band-only records, fixed summation order) -------------------------------------------
// What round 5's sparse merge still paid at 8 ranks on a path that revisits its units (configs[3]: 85 % of the union is multi-toucher): ONE ring
// reduction of 2 MiB per multi-toucher unit from EVERY rank, zeros included, through one link after the other, summed in an order RCCL chooses.
// Here every unit of the union gets an OWNER -- among its touchers the one that observed the most voxels -- and
//   1. every other toucher packs the voxels it observed (weight != 0: the truncation band, ~0.2 of a touched unit) as a RECORD
//      [occupancy bitmap | {sdf, weight} of the observed voxels] and sends it straight to the owner: ONE grouped point-to-point step, every
//      (sender, owner) pair a link of its own;
//   2. the owner adds the records to its own voxels IN RANK ORDER in one kernel: SW = sum_r sdf_r * w_r, W = sum_r w_r, sdf = SW / W -- the
//      algebra of TSDFVolume.cpp:93-94 with a summation order that is a function of the key set alone (bit-reproducible, independent of the wire);
//   3. the non-owners drop their copy.  root == MERGE_DISTRIBUTED stops here: the merged volume stays distributed by owner, every unit complete on
//      exactly one rank (what SaveWorld needs is per-unit, TSDFVolume.cpp:104-132).  root >= 0 / MERGE_ALL: the owners send their finished units
//      -- band records again -- to the root / to everybody in a second grouped step.
// Single-toucher units are their toucher's: they do not move in distributed mode and travel as band records (bit for bit) otherwise.
enum { MERGE_ALL = -1, MERGE_DISTRIBUTED = -2 };

struct OwnerMergeVolume : MergeVolume {
  virtual int band_counts(const int* keys, int n, int* counts) = 0;              // observed voxels (weight != 0) per unit this rank holds
  virtual size_t band_record_floats(int count) const = 0;                         // size of a record with `count` observed voxels
  // records of the given units back to back in one block (the memory space the transport moves); counts as band_counts returned them
  virtual int export_band(const int* keys, const int* counts, int n, float** block) = 0;
  virtual int band_receive_buffer(size_t floats, int which, float** block) = 0;   // which = 0 / 1: the two steps keep their buffers apart
  // owner step: unit keys[u] <- rank-ordered sum over its sources; src[u] lists the records of the OTHER touchers in rank order, self_pos[u] =
  // how many of them come before this rank's own voxels
  virtual int merge_band(const int* keys, int n, const std::vector<std::vector<const float*>>& src, const int* self_pos) = 0;
  virtual int import_band(const int* keys, int n, const std::vector<const float*>& recs) = 0;   // create / overwrite, bit for bit
  virtual int drop_units(const int* keys, int n) = 0;
};

struct OwnerMergeStats {
  int union_units = 0, multi_units = 0, single_units = 0;
  int owned_units = 0;          // units this rank owns after step 2 (multi-toucher units it won + its single-toucher units)
  int owned_multi = 0;          // ... of which it summed
  int dropped_units = 0;        // multi-toucher units this rank touched and handed over
  size_t sent_floats[2] = {0, 0}, received_floats[2] = {0, 0};   // step 1 (records to the owners) and step 3 (finished units to the root / everybody)
  size_t dense_floats = 0;      // what the ring reduction of round 5 would have been handed for the same key sets: 2 x voxels x multi-toucher units
};

namespace detail {
inline unsigned mix_key(int key) {
  unsigned x = (unsigned)key * 2654435761u;
  return x ^ (x >> 15);
}
}  // namespace detail

inline int merge_protocol_owner(MergeTransport& t, OwnerMergeVolume& v, int root, int* union_units, int pre_status = 0, OwnerMergeStats* stats = nullptr) {
  if (union_units) *union_units = 0;
  OwnerMergeStats st;
  std::vector<int> keys, counts;
  int st1 = (pre_status || v.touched_keys(keys)) ? 1 : 0;
  if (!st1) {
    counts.assign(keys.size(), 0);
    if (!keys.empty() && v.band_counts(keys.data(), (int)keys.size(), counts.data())) st1 = 1;
  }
  if (st1) keys.clear(), counts.clear();
  int agree[2] = {(int)keys.size(), st1};
  if (t.allreduce_max(agree, 2)) return MERGE_TRANSPORT_FAILURE;
  if (agree[1]) return st1 ? MERGE_LOCAL_FAILURE : MERGE_PEER_FAILURE;
  const int max_keys = agree[0];
  if (max_keys <= 0) {
    if (stats) *stats = st;
    return MERGE_OK;
  }
  const int W = t.world(), me = t.rank();
  // [keys padded with -1 | counts] of every rank
  std::vector<int> padded((size_t)max_keys * 2, -1), all((size_t)max_keys * 2 * (size_t)W, -1);
  std::copy(keys.begin(), keys.end(), padded.begin());
  std::copy(counts.begin(), counts.end(), padded.begin() + max_keys);
  if (t.allgather(padded.data(), 2 * max_keys, all.data())) return MERGE_TRANSPORT_FAILURE;
  struct Touch { int key, rank, count; };
  std::vector<Touch> kr;
  for (int q = 0; q < W; q++) {
    const int* seg = all.data() + (size_t)q * 2 * max_keys;
    for (int i = 0; i < max_keys; i++)
      if (seg[i] >= 0) kr.push_back(Touch{seg[i], q, std::max(seg[max_keys + i], 0)});
  }
  std::sort(kr.begin(), kr.end(), [](const Touch& a, const Touch& b) { return a.key != b.key ? a.key < b.key : a.rank < b.rank; });
  kr.erase(std::unique(kr.begin(), kr.end(), [](const Touch& a, const Touch& b) { return a.key == b.key && a.rank == b.rank; }), kr.end());
  // the plan: a function of the gathered (key, rank, count) triples alone -- identical on every rank
  struct Unit { int key, owner, first, ntouch; };                      // touchers = kr[first .. first + ntouch), ascending rank
  std::vector<Unit> units;
  for (size_t i = 0; i < kr.size();) {
    size_t j = i;
    while (j < kr.size() && kr[j].key == kr[i].key) j++;
    int best = 0, ntied = 0;
    for (size_t k = i; k < j; k++) best = std::max(best, kr[k].count);
    for (size_t k = i; k < j; k++) ntied += kr[k].count == best ? 1 : 0;
    int pick = (int)(detail::mix_key(kr[i].key) % (unsigned)ntied), owner = kr[i].rank;   // equal counts (a revolution every rank repeats): spread by key
    for (size_t k = i; k < j; k++)
      if (kr[k].count == best && pick-- == 0) owner = kr[k].rank;
    units.push_back(Unit{kr[i].key, owner, (int)i, (int)(j - i)});
    i = j;
  }
  st.union_units = (int)units.size();
  if (union_units) *union_units = st.union_units;
  const size_t unit_floats = 2 * v.unit_voxels();
  // step 1: what I send to each owner, what I receive as an owner
  std::vector<int> send_keys, send_counts, owned_multi, dropped;       // send_*: ordered by (owner, key)
  std::vector<size_t> send_off((size_t)W, 0), send_cnt((size_t)W, 0), recv_cnt((size_t)W, 0);
  std::vector<std::vector<size_t>> rec_floats_from((size_t)W);         // sizes of the records I receive, by sender, in key order
  for (int q = 0; q < W; q++) {
    send_off[(size_t)q] = 0;
    for (size_t d = 0; d < (size_t)q; d++) send_off[(size_t)q] += send_cnt[d];
    for (const Unit& u : units) {
      if (u.ntouch < 2) continue;
      for (int k = 0; k < u.ntouch; k++) {
        const Touch& c = kr[(size_t)(u.first + k)];
        if (c.rank == me && u.owner == q && q != me) {                 // mine, owned by q: goes out
          send_keys.push_back(u.key);
          send_counts.push_back(c.count);
          send_cnt[(size_t)q] += v.band_record_floats(c.count);
        }
        if (u.owner == me && c.rank == q && q != me) {                 // q's, owned by me: comes in
          rec_floats_from[(size_t)q].push_back(v.band_record_floats(c.count));
          recv_cnt[(size_t)q] += rec_floats_from[(size_t)q].back();
        }
      }
    }
  }
  for (const Unit& u : units) {
    const bool mine = std::any_of(kr.begin() + u.first, kr.begin() + u.first + u.ntouch, [&](const Touch& c) { return c.rank == me; });
    if (u.ntouch >= 2) {
      st.multi_units++;
      if (u.owner == me) owned_multi.push_back(u.key);
      else if (mine) dropped.push_back(u.key);
    } else {
      st.single_units++;
    }
    if (u.owner == me) st.owned_units++;
  }
  st.owned_multi = (int)owned_multi.size();
  st.dropped_units = (int)dropped.size();
  st.dense_floats = (size_t)st.multi_units * unit_floats;
  size_t recv_total = 0;
  for (size_t n : recv_cnt) recv_total += n;
  for (size_t n : send_cnt) st.sent_floats[0] += n;
  st.received_floats[0] = recv_total;
  float *send = nullptr, *recv = nullptr;
  int st2 = 0;
  if (!send_keys.empty() && v.export_band(send_keys.data(), send_counts.data(), (int)send_keys.size(), &send)) st2 = 1;
  if (!st2 && recv_total && v.band_receive_buffer(recv_total, 0, &recv)) st2 = 1;
  int any = st2;
  if (t.allreduce_max(&any, 1)) return MERGE_TRANSPORT_FAILURE;
  if (any) return st2 ? MERGE_LOCAL_FAILURE : MERGE_PEER_FAILURE;
  if (st.multi_units > 0 && W > 1) {                                   // (the same decision on every rank)
    if (t.exchange_v(send, send_off, send_cnt, recv, recv_cnt)) return MERGE_TRANSPORT_FAILURE;
  }
  // step 2: the owner's rank-ordered sums.  recv holds the senders' blocks in rank order, each block its records in key order.
  int st3 = 0;
  if (!owned_multi.empty()) {
    std::vector<size_t> cursor((size_t)W, 0), block_off((size_t)W, 0), taken((size_t)W, 0);
    for (int q = 1; q < W; q++) block_off[(size_t)q] = block_off[(size_t)q - 1] + recv_cnt[(size_t)q - 1];
    std::vector<std::vector<const float*>> src;
    std::vector<int> self_pos;
    for (const Unit& u : units) {
      if (u.ntouch < 2 || u.owner != me) continue;
      std::vector<const float*> s;
      int pos = 0;
      for (int k = 0; k < u.ntouch; k++) {
        const int q = kr[(size_t)(u.first + k)].rank;
        if (q == me) { pos = (int)s.size(); continue; }
        s.push_back(recv + block_off[(size_t)q] + cursor[(size_t)q]);
        cursor[(size_t)q] += rec_floats_from[(size_t)q][taken[(size_t)q]++];
      }
      src.push_back(s);
      self_pos.push_back(pos);
    }
    if (v.merge_band(owned_multi.data(), (int)owned_multi.size(), src, self_pos.data())) st3 = 1;
  }
  if (!st3 && !dropped.empty() && v.drop_units(dropped.data(), (int)dropped.size())) st3 = 1;
  if (root == MERGE_DISTRIBUTED || W == 1) {
    any = st3;                                                         // everybody learns whether the merged volume is whole
    if (t.allreduce_max(&any, 1)) return MERGE_TRANSPORT_FAILURE;
    if (stats) *stats = st;
    return any ? (st3 ? MERGE_LOCAL_FAILURE : MERGE_PEER_FAILURE) : MERGE_OK;
  }
  // step 3: the finished units travel to the root / to everybody.  The band of a summed unit is the union of its touchers' bands: one more
  // small all-gather ([status | count per multi-toucher unit, -1 where this rank is not the owner]).
  std::vector<int> fin_counts(owned_multi.size(), 0);
  if (!st3 && !owned_multi.empty() && v.band_counts(owned_multi.data(), (int)owned_multi.size(), fin_counts.data())) st3 = 1;
  std::vector<int> mine2((size_t)st.multi_units + 1, -1), all2(((size_t)st.multi_units + 1) * (size_t)W, -1);
  mine2[0] = st3;
  {
    size_t m = 0, o = 0;
    for (const Unit& u : units) {
      if (u.ntouch < 2) continue;
      if (u.owner == me) mine2[1 + m] = fin_counts[o++];
      m++;
    }
  }
  if (t.allgather(mine2.data(), st.multi_units + 1, all2.data())) return MERGE_TRANSPORT_FAILURE;
  for (int q = 0; q < W; q++)
    if (all2[(size_t)q * ((size_t)st.multi_units + 1)] > 0) return st3 ? MERGE_LOCAL_FAILURE : MERGE_PEER_FAILURE;
  // every unit with its owner and its final count, key order
  std::vector<int> out_keys, out_counts;                               // what I own: ONE block, the same for every receiver
  std::vector<std::vector<int>> in_keys((size_t)W);
  std::vector<std::vector<size_t>> in_floats((size_t)W);
  {
    size_t m = 0;
    for (const Unit& u : units) {
      int cnt;
      if (u.ntouch >= 2) {
        cnt = all2[(size_t)u.owner * ((size_t)st.multi_units + 1) + 1 + m];
        m++;
      } else {
        cnt = kr[(size_t)u.first].count;
      }
      if (u.owner == me) {
        out_keys.push_back(u.key);
        out_counts.push_back(cnt);
      } else if (root == MERGE_ALL || root == me) {
        in_keys[(size_t)u.owner].push_back(u.key);
        in_floats[(size_t)u.owner].push_back(v.band_record_floats(cnt));
      }
    }
  }
  const bool i_send = !out_keys.empty() && (root == MERGE_ALL || root != me);
  size_t out_floats = 0;
  for (int c : out_counts) out_floats += v.band_record_floats(c);
  std::vector<size_t> off2((size_t)W, 0), cnt2((size_t)W, 0), rcv2((size_t)W, 0);
  for (int q = 0; q < W; q++) {
    if (i_send && q != me && (root == MERGE_ALL || q == root)) cnt2[(size_t)q] = out_floats;
    for (size_t n : in_floats[(size_t)q]) rcv2[(size_t)q] += n;
  }
  size_t rcv_total = 0;
  for (size_t n : rcv2) rcv_total += n;
  float *send2 = nullptr, *recv2 = nullptr;
  int st4 = 0;
  if (i_send && v.export_band(out_keys.data(), out_counts.data(), (int)out_keys.size(), &send2)) st4 = 1;
  if (!st4 && rcv_total && v.band_receive_buffer(rcv_total, 1, &recv2)) st4 = 1;
  any = st4;
  if (t.allreduce_max(&any, 1)) return MERGE_TRANSPORT_FAILURE;
  if (any) return st4 ? MERGE_LOCAL_FAILURE : MERGE_PEER_FAILURE;
  for (size_t n : cnt2) st.sent_floats[1] += n;
  st.received_floats[1] = rcv_total;
  if (t.exchange_v(send2, off2, cnt2, recv2, rcv2)) return MERGE_TRANSPORT_FAILURE;
  if (stats) *stats = st;
  if (rcv_total) {                                                     // after the last collective: nobody waits for this rank
    std::vector<int> ik;
    std::vector<const float*> recs;
    size_t off = 0;
    for (int q = 0; q < W; q++)
      for (size_t i = 0; i < in_keys[(size_t)q].size(); i++) {
        ik.push_back(in_keys[(size_t)q][i]);
        recs.push_back(recv2 + off);
        off += in_floats[(size_t)q][i];
      }
    if (v.import_band(ik.data(), (int)ik.size(), recs)) return MERGE_LOCAL_FAILURE;
  }
  return MERGE_OK;
}

}  // namespace er
