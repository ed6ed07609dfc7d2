// merge_protocol_check.cpp -- CPU test of elasticreconstruction_amd/csrc/er_merge_protocol.h, the protocol behind
// er_tsdf_allreduce: the SAME header that er_multi.hip runs over RCCL runs here over host threads with a shared-memory
// transport and host-array volumes (tiny units), world = 1, 2, 3.  Cases: uneven key counts, an empty rank, every rank empty,
// root = 0 / 1 / 2 / all-reduce (root < 0), a rank whose key query fails (unit pool overflow) and a rank whose export fails:
// in the failure cases EVERY rank must come back nonzero -- none may wait in a collective for ever (the run itself is the hang test,
// tests/test_distributed_cpu.py gives it a timeout).  Round 5 (the sparse merge): the floats handed to the sum reduction must be exactly
// 2 x unit voxels x (units two or more ranks touched), every unit only ONE rank touched must arrive on the receiving rank(s) BIT FOR BIT (it
// travels point to point, no sdf * w / w round trip) and must not move at all when it already lives where it is wanted, and the merged volume
// must equal the dense algebra (w = sum w_g, sdf = sum sdf_g w_g / w) the protocol of rounds 2-4 computed.  Round 6 (the owner merge,
// merge_protocol_owner): the same cases once more through the reduce-scatter by unit -- every unit of the union ends up complete on exactly one rank
// (root = MERGE_DISTRIBUTED), on the root, or on every rank; a multi-toucher unit equals the float32 sum of its touchers IN RANK ORDER bit for bit
// (the order is a function of the key sets), a single-toucher unit is untouched; the floats that cross the transport are exactly the band records of
// the non-owning touchers (+ the finished units on their way to the root), never a whole plane.  Prints "OK <cases>" and exits 0.
#include "er_merge_protocol.h"

#include <cmath>
#include <cstring>
#include <set>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <thread>

namespace {

const size_t VOX = 8;   // voxels per unit in this test

struct Shared {          // one per communicator
  int world;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0, generation = 0;
  std::vector<const int*> iptr;
  std::vector<float*> fptr;
  std::vector<int> ibuf;
  std::vector<float> fbuf;
  std::vector<const float*> xsend;                       // exchange: every rank's send block, its length and its receivers
  std::vector<size_t> xcount;
  std::vector<std::vector<int>> xto;
  std::vector<std::vector<size_t>> voff, vcnt;           // exchange_v
  size_t moved_v_floats = 0;
  int exchange_v_calls = 0;
  size_t reduced_floats = 0, moved_floats = 0;           // what the data-path steps were handed (reduce: per call; exchange: per (sender, receiver))
  int reduce_calls = 0, exchange_calls = 0;
  explicit Shared(int w) : world(w), iptr((size_t)w), fptr((size_t)w), xsend((size_t)w), xcount((size_t)w), xto((size_t)w), voff((size_t)w), vcnt((size_t)w) {}
  // classic generation barrier; `last` runs inside the critical section of the last arriver
  template <class F> void barrier(F last) {
    std::unique_lock<std::mutex> lk(m);
    const int gen = generation;
    if (++arrived == world) {
      last();
      arrived = 0;
      generation++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
  }
};

struct ThreadTransport : er::MergeTransport {
  Shared& s;
  int r;
  ThreadTransport(Shared& sh, int rank_) : s(sh), r(rank_) {}
  int rank() const override { return r; }
  int world() const override { return s.world; }
  int allreduce_max(int* v, int n) override {
    s.iptr[(size_t)r] = v;
    s.barrier([&] {
      s.ibuf.assign((size_t)n, -2147483647 - 1);
      for (int q = 0; q < s.world; q++)
        for (int i = 0; i < n; i++) s.ibuf[(size_t)i] = std::max(s.ibuf[(size_t)i], s.iptr[(size_t)q][i]);
    });
    for (int i = 0; i < n; i++) v[i] = s.ibuf[(size_t)i];
    s.barrier([] {});                                    // nobody overwrites ibuf before everyone has read it
    return 0;
  }
  int allgather(const int* mine, int n, int* all) override {
    s.iptr[(size_t)r] = mine;
    s.barrier([&] {
      s.ibuf.resize((size_t)n * s.world);
      for (int q = 0; q < s.world; q++) std::copy(s.iptr[(size_t)q], s.iptr[(size_t)q] + n, s.ibuf.begin() + (size_t)q * n);
    });
    std::copy(s.ibuf.begin(), s.ibuf.end(), all);
    s.barrier([] {});
    return 0;
  }
  int reduce_sum(float* planes, size_t count, int root) override {
    s.fptr[(size_t)r] = planes;
    s.barrier([&] {
      s.fbuf.assign(count, 0.f);
      for (int q = 0; q < s.world; q++)                  // rank order: a fixed summation order, like a ring would give
        for (size_t i = 0; i < count; i++) s.fbuf[i] += s.fptr[(size_t)q][i];
      s.reduced_floats += count;
      s.reduce_calls++;
    });
    if (root < 0 || root == r) std::copy(s.fbuf.begin(), s.fbuf.end(), planes);
    s.barrier([] {});
    return 0;
  }
  int exchange(const float* send, size_t send_count, const std::vector<int>& send_to, float* recv, const std::vector<size_t>& recv_count) override {
    s.xsend[(size_t)r] = send;
    s.xcount[(size_t)r] = send_count;
    s.xto[(size_t)r] = send_to;
    s.barrier([&] {
      s.exchange_calls++;
      for (int q = 0; q < s.world; q++) s.moved_floats += s.xcount[(size_t)q] * s.xto[(size_t)q].size();
    });
    size_t off = 0;
    int bad = 0;
    for (int q = 0; q < s.world; q++) {
      const size_t n = recv_count[(size_t)q];
      if (!n) continue;
      const std::vector<int>& to = s.xto[(size_t)q];     // what I expect from q must be what q sends to me
      if (q == r || n != s.xcount[(size_t)q] || std::find(to.begin(), to.end(), r) == to.end()) bad = 1;
      else std::copy(s.xsend[(size_t)q], s.xsend[(size_t)q] + n, recv + off);
      off += n;
    }
    for (int q : send_to)                                // ... and whoever I send to must expect exactly that (checked from the sender's side by the receiver above)
      if (q == r || q < 0 || q >= s.world) bad = 1;
    s.barrier([] {});
    return bad;
  }
  int exchange_v(const float* send, const std::vector<size_t>& send_off, const std::vector<size_t>& send_count, float* recv,
                 const std::vector<size_t>& recv_count) override {
    s.xsend[(size_t)r] = send;
    s.voff[(size_t)r] = send_off;
    s.vcnt[(size_t)r] = send_count;
    s.barrier([&] {
      s.exchange_v_calls++;
      for (int q = 0; q < s.world; q++)
        for (size_t n : s.vcnt[(size_t)q]) s.moved_v_floats += n;
    });
    size_t off = 0;
    int bad = 0;
    for (int q = 0; q < s.world; q++) {
      const size_t n = recv_count[(size_t)q];
      if (s.vcnt[(size_t)q][(size_t)r] != n || (q == r && n)) { bad = 1; continue; }
      if (!n) continue;
      std::copy(s.xsend[(size_t)q] + s.voff[(size_t)q][(size_t)r], s.xsend[(size_t)q] + s.voff[(size_t)q][(size_t)r] + n, recv + off);
      off += n;
    }
    s.barrier([] {});
    return bad;
  }
};

struct HostVolume : er::OwnerMergeVolume {
  std::map<int, std::vector<float>> sdf, w;              // key -> VOX values
  std::vector<float> planes, raw_out, raw_in;
  bool fail_keys = false, fail_export = false;
  size_t unit_voxels() const override { return VOX; }
  int touched_keys(std::vector<int>& keys) override {
    if (fail_keys) return 1;
    keys.clear();
    for (auto& kv : sdf) keys.push_back(kv.first);
    return 0;
  }
  int export_planes(const int* uk, int nu, float** out) override {
    if (fail_export) return 1;
    planes.assign((size_t)nu * 2 * VOX, 0.f);
    for (int u = 0; u < nu; u++) {
      auto it = sdf.find(uk[u]);
      if (it == sdf.end()) continue;                     // a unit this rank never touched contributes zeros
      for (size_t i = 0; i < VOX; i++) {
        planes[((size_t)u * 2) * VOX + i] = it->second[i] * w[uk[u]][i];
        planes[((size_t)u * 2 + 1) * VOX + i] = w[uk[u]][i];
      }
    }
    *out = planes.data();
    return 0;
  }
  int export_raw(const int* uk, int nu, float** out) override {
    if (fail_export) return 1;
    raw_out.assign((size_t)nu * 2 * VOX, 0.f);
    for (int u = 0; u < nu; u++) {
      if (!sdf.count(uk[u])) return 1;                    // the protocol only asks for units this rank owns
      std::copy(sdf[uk[u]].begin(), sdf[uk[u]].end(), raw_out.begin() + ((size_t)u * 2) * VOX);
      std::copy(w[uk[u]].begin(), w[uk[u]].end(), raw_out.begin() + ((size_t)u * 2 + 1) * VOX);
    }
    *out = raw_out.data();
    return 0;
  }
  int receive_buffer(int nu, float** out) override {
    raw_in.assign((size_t)nu * 2 * VOX, -7.f);
    *out = raw_in.data();
    return 0;
  }
  int import_raw(const int* uk, int nu, const float* p) override {
    for (int u = 0; u < nu; u++) {
      sdf[uk[u]].assign(p + ((size_t)u * 2) * VOX, p + ((size_t)u * 2 + 1) * VOX);
      w[uk[u]].assign(p + ((size_t)u * 2 + 1) * VOX, p + ((size_t)u * 2 + 2) * VOX);
    }
    return 0;
  }
  // ---- owner merge: a record = [bitmap word (as float bits)] + {sdf, w} of the observed voxels (VOX <= 32 here) ----
  std::vector<float> band_out, band_in[2];
  int band_counts(const int* uk, int nu, int* counts) override {
    for (int u = 0; u < nu; u++) {
      if (!w.count(uk[u])) return 1;
      int c = 0;
      for (float x : w[uk[u]]) c += x != 0.f;
      counts[u] = c;
    }
    return 0;
  }
  size_t band_record_floats(int count) const override { return 1 + 2 * (size_t)count; }
  int export_band(const int* uk, const int* counts, int nu, float** out) override {
    if (fail_export) return 1;
    band_out.clear();
    for (int u = 0; u < nu; u++) {
      if (!w.count(uk[u])) return 1;
      unsigned bits = 0;
      std::vector<float> vals;
      for (size_t i = 0; i < VOX; i++)
        if (w[uk[u]][i] != 0.f) {
          bits |= 1u << i;
          vals.push_back(sdf[uk[u]][i]);
          vals.push_back(w[uk[u]][i]);
        }
      if ((int)vals.size() != 2 * counts[u]) return 1;
      float fb;
      memcpy(&fb, &bits, 4);
      band_out.push_back(fb);
      band_out.insert(band_out.end(), vals.begin(), vals.end());
    }
    *out = band_out.data();
    return 0;
  }
  int band_receive_buffer(size_t floats, int which, float** out) override {
    band_in[which].assign(floats, -7.f);
    *out = band_in[which].data();
    return 0;
  }
  static void unpack(const float* rec, std::vector<float>& S, std::vector<float>& Wt) {
    unsigned bits;
    memcpy(&bits, rec, 4);
    S.assign(VOX, 0.f);
    Wt.assign(VOX, 0.f);
    size_t o = 1;
    for (size_t i = 0; i < VOX; i++)
      if (bits >> i & 1u) {
        S[i] = rec[o];
        Wt[i] = rec[o + 1];
        o += 2;
      }
  }
  int merge_band(const int* uk, int nu, const std::vector<std::vector<const float*>>& src, const int* self_pos) override {
    for (int u = 0; u < nu; u++) {
      if (!sdf.count(uk[u])) return 1;
      std::vector<float> SW(VOX, 0.f), Wt(VOX, 0.f), s1, w1;
      for (int k = 0; k <= (int)src[(size_t)u].size(); k++) {
        if (k == self_pos[u])
          for (size_t i = 0; i < VOX; i++) { SW[i] += sdf[uk[u]][i] * w[uk[u]][i]; Wt[i] += w[uk[u]][i]; }
        if (k == (int)src[(size_t)u].size()) break;
        unpack(src[(size_t)u][(size_t)k], s1, w1);
        for (size_t i = 0; i < VOX; i++)
          if (w1[i] != 0.f) { SW[i] += s1[i] * w1[i]; Wt[i] += w1[i]; }
      }
      for (size_t i = 0; i < VOX; i++) { sdf[uk[u]][i] = Wt[i] > 0.f ? SW[i] / Wt[i] : 0.f; w[uk[u]][i] = Wt[i]; }
    }
    return 0;
  }
  int import_band(const int* uk, int nu, const std::vector<const float*>& recs) override {
    for (int u = 0; u < nu; u++) unpack(recs[(size_t)u], sdf[uk[u]], w[uk[u]]);
    return 0;
  }
  int drop_units(const int* uk, int nu) override {
    for (int u = 0; u < nu; u++) { sdf.erase(uk[u]); w.erase(uk[u]); }
    return 0;
  }
  int import_planes(const int* uk, int nu, const float* p) override {
    for (int u = 0; u < nu; u++) {
      std::vector<float>&S = sdf[uk[u]], &W = w[uk[u]];
      S.assign(VOX, 0.f);
      W.assign(VOX, 0.f);
      for (size_t i = 0; i < VOX; i++) {
        const float ww = p[((size_t)u * 2 + 1) * VOX + i];
        W[i] = ww;
        S[i] = ww > 0.f ? p[((size_t)u * 2) * VOX + i] / ww : 0.f;
      }
    }
    return 0;
  }
};

unsigned rng_state = 12345u;
float frand() { rng_state = rng_state * 1664525u + 1013904223u; return (float)((rng_state >> 8) & 0xffff) / 65536.0f; }

struct Case { int world, root; std::vector<int> nkeys; int fail_keys_rank, fail_export_rank; int pre_status_rank = -1; };

int run_case(const Case& c, int id) {
  const int W = c.world;
  std::vector<HostVolume> vols((size_t)W);
  // expected result of the sequential algebra: w = sum w_g, sdf = sum sdf_g w_g / w  (rank order)
  std::map<int, std::vector<double>> sw, ww;
  for (int r = 0; r < W; r++) {
    for (int k = 0; k < c.nkeys[(size_t)r]; k++) {
      const int key = 1000 + ((k * 7 + r * 3) % 23) * (r % 2 ? 1 : 2);   // overlapping but different key sets per rank
      if (vols[(size_t)r].sdf.count(key)) continue;
      std::vector<float> S(VOX), Wt(VOX);
      for (size_t i = 0; i < VOX; i++) { Wt[i] = (float)(int)(frand() * 40.f); S[i] = Wt[i] > 0 ? frand() * 2.f - 1.f : 0.f; }
      vols[(size_t)r].sdf[key] = S;
      vols[(size_t)r].w[key] = Wt;
    }
    vols[(size_t)r].fail_keys = r == c.fail_keys_rank;
    vols[(size_t)r].fail_export = r == c.fail_export_rank;
  }
  for (int r = 0; r < W; r++)
    for (auto& kv : vols[(size_t)r].sdf) {
      auto& a = sw[kv.first]; auto& b = ww[kv.first];
      a.resize(VOX, 0.0); b.resize(VOX, 0.0);
      for (size_t i = 0; i < VOX; i++) { a[i] += (double)(kv.second[i] * vols[(size_t)r].w[kv.first][i]); b[i] += vols[(size_t)r].w[kv.first][i]; }
    }
  std::vector<HostVolume> before = vols;
  Shared sh(W);
  std::vector<int> rc((size_t)W, -1), nu((size_t)W, -1);
  std::vector<er::MergeStats> stats((size_t)W);
  std::vector<std::thread> th;
  for (int r = 0; r < W; r++)
    th.emplace_back([&, r] {
      ThreadTransport t(sh, r);
      rc[(size_t)r] = er::merge_protocol(t, vols[(size_t)r], c.root, &nu[(size_t)r], r == c.pre_status_rank ? 1 : 0, &stats[(size_t)r]);
    });
  for (auto& t : th) t.join();
  // who touched what, from the volumes as they were before the merge
  std::map<int, std::vector<int>> touchers;
  for (int r = 0; r < W; r++)
    for (auto& kv : before[(size_t)r].sdf) touchers[kv.first].push_back(r);
  size_t n_multi = 0, n_travel = 0, n_pairs_moved = 0;
  for (auto& kv : touchers) {
    if (kv.second.size() >= 2) { n_multi++; continue; }
    const bool travels = c.root < 0 ? W > 1 : kv.second[0] != c.root;
    if (travels) { n_travel++; n_pairs_moved += c.root < 0 ? (size_t)(W - 1) : 1; }
  }
  const bool expect_fail = c.fail_keys_rank >= 0 || c.fail_export_rank >= 0 || c.pre_status_rank >= 0;
  for (int r = 0; r < W; r++) {
    if (expect_fail) {
      const bool me = r == c.fail_keys_rank || r == c.fail_export_rank || r == c.pre_status_rank;
      const int want = me ? er::MERGE_LOCAL_FAILURE : er::MERGE_PEER_FAILURE;
      // with only an export failure armed the union may be empty (nothing touched): then nobody fails -- not a case we build
      if (rc[(size_t)r] != want) { fprintf(stderr, "case %d rank %d: rc %d, want %d\n", id, r, rc[(size_t)r], want); return 1; }
      // nothing was merged anywhere
      if (vols[(size_t)r].sdf != before[(size_t)r].sdf || vols[(size_t)r].w != before[(size_t)r].w) { fprintf(stderr, "case %d rank %d: volume changed by a failed merge\n", id, r); return 1; }
      continue;
    }
    if (rc[(size_t)r] != er::MERGE_OK || nu[(size_t)r] != (int)sw.size()) { fprintf(stderr, "case %d rank %d: rc %d union %d want %zu\n", id, r, rc[(size_t)r], nu[(size_t)r], sw.size()); return 1; }
    // the sparse merge: ONLY multi-toucher units go through the sum reduction (one call), single-toucher units travel point to point or not at all
    if (stats[(size_t)r].multi_units != (int)n_multi || stats[(size_t)r].reduced_floats != n_multi * 2 * VOX ||
        stats[(size_t)r].single_units != (int)(touchers.size() - n_multi)) {
      fprintf(stderr, "case %d rank %d: stats multi %d reduced %zu single %d, want %zu / %zu / %zu\n", id, r, stats[(size_t)r].multi_units,
              stats[(size_t)r].reduced_floats, stats[(size_t)r].single_units, n_multi, n_multi * 2 * VOX, touchers.size() - n_multi);
      return 1;
    }
    if (r == 0 && (sh.reduced_floats != n_multi * 2 * VOX || sh.reduce_calls != (n_multi ? 1 : 0) || sh.moved_floats != n_pairs_moved * 2 * VOX ||
                   sh.exchange_calls != (n_travel ? 1 : 0))) {
      fprintf(stderr, "case %d: transport saw %zu reduced floats in %d calls, %zu moved floats in %d exchanges; want %zu / %d / %zu / %d\n", id, sh.reduced_floats,
              sh.reduce_calls, sh.moved_floats, sh.exchange_calls, n_multi * 2 * VOX, n_multi ? 1 : 0, n_pairs_moved * 2 * VOX, n_travel ? 1 : 0);
      return 1;
    }
    const bool receives = c.root < 0 || c.root == r;
    if (!receives) {
      if (vols[(size_t)r].sdf != before[(size_t)r].sdf) { fprintf(stderr, "case %d rank %d: non-root volume changed\n", id, r); return 1; }
      continue;
    }
    if (vols[(size_t)r].sdf.size() != sw.size()) { fprintf(stderr, "case %d rank %d: %zu units after the merge, want %zu\n", id, r, vols[(size_t)r].sdf.size(), sw.size()); return 1; }
    for (auto& kv : touchers)                              // a unit only one rank touched arrives (or stays) exactly as its owner had it
      if (kv.second.size() == 1) {
        const HostVolume& o = before[(size_t)kv.second[0]];
        if (vols[(size_t)r].sdf[kv.first] != o.sdf.at(kv.first) || vols[(size_t)r].w[kv.first] != o.w.at(kv.first)) {
          fprintf(stderr, "case %d rank %d key %d: a single-toucher unit changed on its way\n", id, r, kv.first);
          return 1;
        }
      }
    for (auto& kv : sw)
      for (size_t i = 0; i < VOX; i++) {
        const double Wd = ww[kv.first][i], Sd = Wd > 0 ? kv.second[i] / Wd : 0.0;
        if ((double)vols[(size_t)r].w[kv.first][i] != Wd) { fprintf(stderr, "case %d rank %d key %d: weight %g want %g\n", id, r, kv.first, vols[(size_t)r].w[kv.first][i], Wd); return 1; }
        if (std::fabs((double)vols[(size_t)r].sdf[kv.first][i] - Sd) > 1e-5) { fprintf(stderr, "case %d rank %d key %d: sdf off by %g\n", id, r, kv.first, std::fabs(vols[(size_t)r].sdf[kv.first][i] - Sd)); return 1; }
      }
  }
  return 0;
}


// The same volumes through merge_protocol_owner.  root: >= 0, er::MERGE_ALL or er::MERGE_DISTRIBUTED.
int run_owner_case(const Case& c, int id, int root) {
  const int W = c.world;
  std::vector<HostVolume> vols((size_t)W);
  rng_state = 777u + (unsigned)id;
  for (int r = 0; r < W; r++) {
    for (int k = 0; k < c.nkeys[(size_t)r]; k++) {
      const int key = 1000 + ((k * 7 + r * 3) % 23) * (r % 2 ? 1 : 2);
      if (vols[(size_t)r].sdf.count(key)) continue;
      std::vector<float> S(VOX), Wt(VOX);
      for (size_t i = 0; i < VOX; i++) { Wt[i] = frand() < 0.4f ? 0.f : (float)(1 + (int)(frand() * 40.f)); S[i] = Wt[i] > 0 ? frand() * 2.f - 1.f : 0.f; }
      vols[(size_t)r].sdf[key] = S;
      vols[(size_t)r].w[key] = Wt;
    }
    vols[(size_t)r].fail_keys = r == c.fail_keys_rank;
    vols[(size_t)r].fail_export = r == c.fail_export_rank;
  }
  const std::vector<HostVolume> before = vols;
  std::map<int, std::vector<int>> touchers;
  for (int r = 0; r < W; r++)
    for (auto& kv : before[(size_t)r].sdf) touchers[kv.first].push_back(r);
  // expected: float32 sums in RANK order (what the owner computes, bit for bit), and the double algebra within 1e-5
  std::map<int, std::vector<float>> eS, eW;
  size_t n_multi = 0;
  for (auto& kv : touchers) {
    std::vector<float> SW(VOX, 0.f), Wt(VOX, 0.f);
    if (kv.second.size() == 1) {
      eS[kv.first] = before[(size_t)kv.second[0]].sdf.at(kv.first);
      eW[kv.first] = before[(size_t)kv.second[0]].w.at(kv.first);
      continue;
    }
    n_multi++;
    for (int r : kv.second)
      for (size_t i = 0; i < VOX; i++) {
        const float s1 = before[(size_t)r].sdf.at(kv.first)[i], w1 = before[(size_t)r].w.at(kv.first)[i];
        if (w1 != 0.f) { SW[i] += s1 * w1; Wt[i] += w1; }
      }
    eS[kv.first].resize(VOX);
    for (size_t i = 0; i < VOX; i++) eS[kv.first][i] = Wt[i] > 0.f ? SW[i] / Wt[i] : 0.f;
    eW[kv.first] = Wt;
  }
  Shared sh(W);
  std::vector<int> rc((size_t)W, -1), nu((size_t)W, -1);
  std::vector<er::OwnerMergeStats> stats((size_t)W);
  std::vector<std::thread> th;
  for (int r = 0; r < W; r++)
    th.emplace_back([&, r] {
      ThreadTransport t(sh, r);
      rc[(size_t)r] = er::merge_protocol_owner(t, vols[(size_t)r], root, &nu[(size_t)r], r == c.pre_status_rank ? 1 : 0, &stats[(size_t)r]);
    });
  for (auto& t : th) t.join();
  const bool expect_fail = c.fail_keys_rank >= 0 || c.fail_export_rank >= 0 || c.pre_status_rank >= 0;
  if (expect_fail) {
    // (an export failure only bites if that rank has something to export: a non-owning toucher, or -- with a root -- an owner that must send)
    bool anyfail = false;
    for (int r = 0; r < W; r++) anyfail = anyfail || rc[(size_t)r] != er::MERGE_OK;
    for (int r = 0; r < W; r++) {
      const bool me = r == c.fail_keys_rank || r == c.fail_export_rank || r == c.pre_status_rank;
      if (anyfail && rc[(size_t)r] != (me ? er::MERGE_LOCAL_FAILURE : er::MERGE_PEER_FAILURE)) {
        fprintf(stderr, "owner case %d root %d rank %d: rc %d\n", id, root, r, rc[(size_t)r]);
        return 1;
      }
    }
    if (!anyfail && c.fail_export_rank < 0) { fprintf(stderr, "owner case %d root %d: a failure went unnoticed\n", id, root); return 1; }
    return 0;
  }
  size_t sent0 = 0, recv0 = 0, sent1 = 0, recv1 = 0;
  std::map<int, int> holders;
  for (int r = 0; r < W; r++) {
    if (rc[(size_t)r] != er::MERGE_OK || nu[(size_t)r] != (int)touchers.size()) { fprintf(stderr, "owner case %d root %d rank %d: rc %d union %d want %zu\n", id, root, r, rc[(size_t)r], nu[(size_t)r], touchers.size()); return 1; }
    const er::OwnerMergeStats& st = stats[(size_t)r];
    if (st.multi_units != (int)n_multi || st.single_units != (int)(touchers.size() - n_multi) || st.dense_floats != n_multi * 2 * VOX) { fprintf(stderr, "owner case %d rank %d: stats\n", id, r); return 1; }
    sent0 += st.sent_floats[0]; recv0 += st.received_floats[0]; sent1 += st.sent_floats[1]; recv1 += st.received_floats[1];
    for (auto& kv : vols[(size_t)r].sdf) holders[kv.first]++;
  }
  if (sent0 != recv0 || sent1 != recv1 || sh.moved_v_floats != sent0 + sent1) { fprintf(stderr, "owner case %d root %d: byte accounting %zu %zu %zu %zu %zu\n", id, root, sent0, recv0, sent1, recv1, sh.moved_v_floats); return 1; }
  // step 1 moves exactly the records of the non-owning touchers: never more than their observed voxels + one bitmap word each
  size_t bound = 0;
  for (auto& kv : touchers)
    if (kv.second.size() >= 2) {
      std::vector<size_t> rec;
      for (int r : kv.second) { int cnt = 0; for (float x : before[(size_t)r].w.at(kv.first)) cnt += x != 0.f; rec.push_back(1 + 2 * (size_t)cnt); }
      size_t tot = 0, mx = 0;
      for (size_t x : rec) { tot += x; mx = std::max(mx, x); }
      bound += tot - mx;                                   // the owner is a toucher with the largest record
    }
  if (sent0 != bound) { fprintf(stderr, "owner case %d root %d: step 1 moved %zu floats, the non-owners' records are %zu\n", id, root, sent0, bound); return 1; }
  for (int r = 0; r < W; r++) {
    const bool all_here = root == er::MERGE_ALL || root == r;
    for (auto& kv : touchers) {
      const bool here = vols[(size_t)r].sdf.count(kv.first) != 0;
      if (all_here && !here) { fprintf(stderr, "owner case %d root %d rank %d: unit %d missing\n", id, root, r, kv.first); return 1; }
      if (!here) continue;
      if (vols[(size_t)r].w[kv.first] != eW[kv.first] || vols[(size_t)r].sdf[kv.first] != eS[kv.first]) { fprintf(stderr, "owner case %d root %d rank %d key %d: not the rank-ordered sum\n", id, root, r, kv.first); return 1; }
    }
  }
  for (auto& kv : touchers) {
    if (root == er::MERGE_DISTRIBUTED && holders[kv.first] != 1) { fprintf(stderr, "owner case %d: unit %d lives on %d ranks after a distributed merge\n", id, kv.first, holders[kv.first]); return 1; }
    if (root == er::MERGE_DISTRIBUTED && kv.second.size() == 1 && !vols[(size_t)kv.second[0]].sdf.count(kv.first)) { fprintf(stderr, "owner case %d: a single-toucher unit moved\n", id); return 1; }
    if (holders[kv.first] < 1) { fprintf(stderr, "owner case %d root %d: unit %d lost\n", id, root, kv.first); return 1; }
  }
  // against the double algebra (the bar of the GPU tests): weights exact, sdf 1e-5
  for (auto& kv : touchers) {
    for (size_t i = 0; i < VOX; i++) {
      double SW = 0, Wd = 0;
      for (int r : kv.second) { SW += (double)(before[(size_t)r].sdf.at(kv.first)[i] * before[(size_t)r].w.at(kv.first)[i]); Wd += before[(size_t)r].w.at(kv.first)[i]; }
      if ((double)eW[kv.first][i] != Wd || std::fabs((double)eS[kv.first][i] - (Wd > 0 ? SW / Wd : 0.0)) > 1e-5) { fprintf(stderr, "owner case %d: expectation off\n", id); return 1; }
    }
  }
  return 0;
}

}  // namespace

int main() {
  const std::vector<Case> cases = {
      {1, 0, {5}, -1, -1},               {2, 0, {7, 3}, -1, -1},          {2, 1, {0, 9}, -1, -1},      // an empty rank, root != 0
      {2, -1, {4, 11}, -1, -1},          {3, 2, {12, 0, 5}, -1, -1},      {3, -1, {1, 2, 20}, -1, -1},
      {3, 0, {0, 0, 0}, -1, -1},         {2, 0, {23, 23}, -1, -1},                                      // nobody touched anything; identical sets
      {2, 0, {6, 6}, 1, -1},             {3, -1, {6, 2, 9}, 0, -1},       {3, 1, {6, 2, 9}, -1, 2},    // key query / export failure on one rank
      {2, -1, {3, 4}, -1, 0},                                                                           // (the failing rank must have something to export: a rank without units exports nothing since round 5)
      {2, 0, {5, 5}, -1, -1, 0},         {3, -1, {4, 0, 7}, -1, -1, 2},                                 // a failure found before the protocol (bad argument on one rank)
      {3, 0, {3, 3, 3}, -1, -1},         {3, 1, {2, 9, 1}, -1, -1},       {2, -1, {1, 1}, -1, -1},     // few keys: mostly single-toucher units, every root
      {3, 2, {0, 0, 8}, -1, -1},         {3, 0, {0, 8, 0}, -1, -1},                                     // everything already on the root / everything has to travel
  };
  int id = 0;
  for (const Case& c : cases)
    if (run_case(c, id++)) return 1;
  int owner_runs = 0;
  id = 0;
  for (const Case& c : cases) {
    std::set<int> roots = {er::MERGE_DISTRIBUTED, er::MERGE_ALL, c.root < 0 ? 0 : c.root, c.world - 1};
    for (int root : roots) {
      if (run_owner_case(c, id, root)) return 1;
      owner_runs++;
    }
    id++;
  }
  printf("OK %zu + %d owner-merge runs\n", cases.size(), owner_runs);
  return 0;
}
