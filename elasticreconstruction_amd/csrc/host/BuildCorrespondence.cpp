// BuildCorrespondence -- drop-in for the reference's BuildCorrespondence.exe
// (BuildCorrespondence/BuildCorrespondence.cpp:10-90, CorresApp.cpp:31-359) with the numeric core
// (transform, exact NN, inlier pre-check, point-to-plane ICP, correspondence filter, information matrix)
// on MI355X through liber_hip.so.
//
// Every reference flag keeps its name, default and meaning:
//   --traj + --num, --interval, --length, --reg_traj, --registration, --reg_dist (also sets
//   dist_thresh_ = reg_dist / 2, BuildCorrespondence.cpp:52-55), --reg_ratio, --reg_num, --blacklist (the
//   help text of the reference says --blasklist but its parser reads --blacklist; both are accepted),
//   --save_xyzn, --output_information, --redux
// Same files: <dir>/cloud_bin_<i>.pcd in; ./reg_output.log, ./reg_output.info, <dir>/corres_<i>_<j>.txt,
// <dir>/cloud_bin_xyzn_<i>.xyzn out.
// Additive: --gpus <n> (1) shards the pair list cyclically over n GPUs of the node (pairs are independent,
// CorresApp.cpp:121,220: no collective), --device <first gpu> (0), --icp_stop_rule 0|1 (PCL 1.7 | <= 1.6).
#include <omp.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <future>
#include <iostream>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include <unistd.h>

#include "../er_mat4.h"
#include "er_formats.h"
#include "er_hip.h"

using erfmt::FramedInformation;
using erfmt::FramedTransformation;

namespace {

int print_help() {
  std::cout << "\nApplication parameters:" << std::endl;
  std::cout << "    --help, -h                      : print this message" << std::endl;
  std::cout << "    --traj <log_file>               : initialization, camera pose trajectory" << std::endl;
  std::cout << "    --num <num_of_fragments>        : use together with --traj" << std::endl;
  std::cout << "    --interval <interval>           : use together with --traj, default : 50" << std::endl;
  std::cout << "    --length <length>               : use together with --traj, default : 3.0" << std::endl;
  std::cout << "    --reg_traj <log_file>           : initialization, registration.log file, will overwrite --traj" << std::endl;
  std::cout << "    --registration                  : registration results are written into reg_output.log file" << std::endl;
  std::cout << "    --reg_dist <dist>               : distance threshold for registration, default 0.03" << std::endl;
  std::cout << "    --reg_ratio <ratio>             : correspondence points are at least <ratio> in each point cloud, default 0.25" << std::endl;
  std::cout << "    --reg_num <number>              : correspondence point number requirement, default 40,000" << std::endl;
  std::cout << "    --blacklist <blacklist_file>    : each line is the block we want to blacklist" << std::endl;
  std::cout << "    --save_xyzn                     : save point cloud into ascii file" << std::endl;
  std::cout << "    --output_information            : output the registration information matrix into reg_output.info" << std::endl;
  std::cout << "    --redux <log_file>              : use transformations in <log_file> as constraints" << std::endl;
  std::cout << "MI355X options:" << std::endl;
  std::cout << "    --gpus <n> (1)  --device <first gpu> (0)  --icp_stop_rule <0: PCL 1.7 | 1: PCL <= 1.6> (0)" << std::endl;
  return 0;
}

// corres_<i>_<j>.txt, CorresApp.cpp:175-184: one "%d %d\n" line per correspondence -- the same bytes as fprintf, formatted by hand into 1 MB
// blocks (fprintf costs ~100 ns per line; a 50-pair list is 10 M lines).
bool write_pair_lines(const char* fn, const int* pl, int nk) {
  FILE* f = fopen(fn, "w");
  if (!f) return false;
  std::vector<char> buf((1u << 20) + 32);
  size_t at = 0;
  auto put = [&](int v) {
    char tmp[12];
    int len = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    do { tmp[len++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) buf[at++] = '-';
    while (len) buf[at++] = tmp[--len];
  };
  bool ok = true;
  for (int q = 0; q < nk; q++) {
    put(pl[2 * (size_t)q]);
    buf[at++] = ' ';
    put(pl[2 * (size_t)q + 1]);
    buf[at++] = '\n';
    if (at >= (1u << 20)) { ok = fwrite(buf.data(), 1, at, f) == at && ok; at = 0; }
  }
  if (at) ok = fwrite(buf.data(), 1, at, f) == at && ok;
  return fclose(f) == 0 && ok;
}

struct HostCloud {
  std::vector<float> xyz, nrm;           // after the NaN-normal filter (CorresApp.cpp:94-98)
  size_t size() const { return xyz.size() / 3; }
};

struct App {
  // CCorresApp members (CorresApp.h:19-44), same names and defaults (CorresApp.cpp:8-24)
  std::vector<FramedTransformation> corres_traj_;
  std::vector<FramedInformation> corres_info_;
  double dist_thresh_ = 0.015, normal_thresh_ = 0.8660;
  bool save_xyzn_ = false, save_corres_ = true;
  double reg_dist_ = 0.03, reg_ratio_ = 0.25;
  int reg_num_ = 40000;
  bool registration_ = false, output_information_ = false;
  std::string m_pDirName;
  std::set<int> blacklist_;
  bool redux_ = false;
  std::vector<FramedTransformation> redux_traj_;
  std::unordered_map<int, int> redux_map_;
  int num_ = 0, interval_ = 50;
  double length_ = 3.0;
  // device side: pointclouds_[gpu][fragment]
  std::vector<HostCloud> host_;
  std::vector<std::vector<er_cloud_t>> pointclouds_;
  int gpus_ = 1, device0_ = 0, stop_rule_ = 0;

  ~App() {
    for (auto& per : pointclouds_)
      for (er_cloud_t c : per) er_cloud_destroy(c);
  }

  double GetVolumeOverlapRatio(const double* trans) const {           // CorresApp.h:64-81
    const int res = 20;
    const double ul = length_ / (double)res;
    int s = 0;
    for (int i = 0; i < res; i++)
      for (int j = 0; j < res; j++)
        for (int k = 0; k < res; k++) {
          const double p[4] = {(i + 0.5) * ul, (j + 0.5) * ul, (k + 0.5) * ul, 1.0};
          double q[3];
          for (int r = 0; r < 3; r++) q[r] = ((trans[r * 4] * p[0] + trans[r * 4 + 1] * p[1]) + trans[r * 4 + 2] * p[2]) + trans[r * 4 + 3] * p[3];
          if (q[0] >= 0 && q[0] <= length_ && q[1] >= 0 && q[1] <= length_ && q[2] >= 0 && q[2] <= length_) s++;
        }
    return (double)s / res / res / res;
  }

  bool LoadData(const std::string& filename, int num) {               // CorresApp.cpp:31-110
    size_t c = filename.rfind('\\');
    if (c == std::string::npos) c = filename.rfind('/');
    m_pDirName = c == std::string::npos ? std::string() : filename.substr(0, c + 1);
    if (num > 0) {
      std::vector<FramedTransformation> temp;
      if (!erfmt::load_log(filename, temp) || (int)temp.size() <= (num - 1) * interval_) {
        fprintf(stderr, "BuildCorrespondence: trajectory %s has too few entries\n", filename.c_str());
        return false;
      }
      corres_traj_.clear();
      num_ = num;
      double basepose[16], baseinverse[16], t0inv[16], leftbase[16];
      er::mat4_identity(basepose);
      basepose[3] = length_ / 2.0; basepose[7] = length_ / 2.0; basepose[11] = -0.3;
      er::mat4_inverse(basepose, baseinverse);
      er::mat4_inverse(temp[0].T, t0inv);
      er::mat4_mul(basepose, t0inv, leftbase);
      std::vector<std::vector<double>> ipose((size_t)num_, std::vector<double>(16));
      for (int i = 0; i < num_; i++) {
        double tmp[16];
        er::mat4_mul(leftbase, temp[(size_t)i * interval_].T, tmp);
        er::mat4_mul(tmp, baseinverse, ipose[(size_t)i].data());
      }
      for (int i = 0; i < num_ - 1; i++) {
        double inv_i[16];
        er::mat4_inverse(ipose[(size_t)i].data(), inv_i);
        FramedTransformation t;
        t.id1 = i; t.id2 = i + 1; t.frame = num_;
        er::mat4_mul(inv_i, ipose[(size_t)i + 1].data(), t.T);
        corres_traj_.push_back(t);
        for (int j = i + 2; j < num; j++) {
          FramedTransformation u;
          u.id1 = i; u.id2 = j; u.frame = num_;
          er::mat4_mul(inv_i, ipose[(size_t)j].data(), u.T);
          if (GetVolumeOverlapRatio(u.T) > 0.3) corres_traj_.push_back(u);
        }
      }
      printf("%d initial matching candidates are created.\n", (int)corres_traj_.size());
    } else {
      if (!erfmt::load_log(filename, corres_traj_) || corres_traj_.empty()) {
        fprintf(stderr, "BuildCorrespondence: cannot read %s\n", filename.c_str());
        return false;
      }
      num_ = corres_traj_[0].frame;
    }
    host_.assign((size_t)num_, HostCloud());
    bool ok = true;
#pragma omp parallel for num_threads(8) schedule(dynamic)
    for (int i = 0; i < num_; i++) {
      char fn[1024];
      snprintf(fn, sizeof fn, "%scloud_bin_%d.pcd", m_pDirName.c_str(), i);
      printf("Load file : %s\n", fn);
      std::vector<std::vector<float>> cols;
      size_t n = 0;
      if (!erfmt::load_pcd_fields(fn, {"x", "y", "z", "normal_x", "normal_y", "normal_z"}, cols, n)) {
        fprintf(stderr, "Error loading file.\n");
#pragma omp atomic write
        ok = false;
        continue;
      }
      HostCloud& h = host_[(size_t)i];
      for (size_t j = 0; j < n; j++)
        if (!std::isnan(cols[3][j])) {                                 // :94-98
          h.xyz.insert(h.xyz.end(), {cols[0][j], cols[1][j], cols[2][j]});
          h.nrm.insert(h.nrm.end(), {cols[3][j], cols[4][j], cols[5][j]});
        }
      if (save_xyzn_) {                                                // :100-108
        snprintf(fn, sizeof fn, "%scloud_bin_xyzn_%d.xyzn", m_pDirName.c_str(), i);
        if (FILE* f = fopen(fn, "w")) {
          for (size_t k = 0; k < h.size(); k++)
            fprintf(f, "%.6f %.6f %.6f %.6f %.6f %.6f\n", h.xyz[3 * k], h.xyz[3 * k + 1], h.xyz[3 * k + 2], h.nrm[3 * k], h.nrm[3 * k + 1], h.nrm[3 * k + 2]);
          fclose(f);
        }
      }
    }
    return ok;
  }

  // Fragments are replicated on every GPU in use (100 x 250k points x 24 B = 600 MB: nothing on 288 GB).
  bool Upload() {
    const float cell = (float)std::max(reg_dist_, dist_thresh_);
    pointclouds_.assign((size_t)gpus_, std::vector<er_cloud_t>((size_t)num_, nullptr));
    std::vector<const float*> xyz((size_t)num_), nrm((size_t)num_);
    std::vector<int> counts((size_t)num_);
    for (int i = 0; i < num_; i++) {
      xyz[(size_t)i] = host_[(size_t)i].xyz.data();
      nrm[(size_t)i] = host_[(size_t)i].nrm.data();
      counts[(size_t)i] = (int)host_[(size_t)i].size();
    }
    for (int g = 0; g < gpus_; g++)                                    // one call per GPU: uploads queued up front, grids built in chunks of 8 clouds
      if (er_cloud_create_batch(num_, xyz.data(), nrm.data(), counts.data(), cell, device0_ + g, pointclouds_[(size_t)g].data()) != 0) {
        fprintf(stderr, "BuildCorrespondence: %s\n", er_last_error());
        return false;
      }
    return true;
  }

  void Blacklist(const std::string& filename) {                        // CorresApp.cpp:330-346
    blacklist_.clear();
    if (FILE* f = fopen(filename.c_str(), "r")) {
      char buf[1024];
      int id;
      while (fgets(buf, 1024, f))
        if (strlen(buf) > 0 && buf[0] != '#' && sscanf(buf, "%d", &id) == 1) blacklist_.insert(id);
      fclose(f);
    }
  }

  int GetReduxIndex(int i, int j) const { return i + j * num_; }       // CorresApp.h:61-63

  void Redux(const std::string& filename) {                            // CorresApp.cpp:348-359
    redux_ = true;
    redux_map_.clear();
    erfmt::load_log(filename, redux_traj_);
    for (int i = 0; i < (int)redux_traj_.size(); i++) redux_map_.insert({GetReduxIndex(redux_traj_[(size_t)i].id1, redux_traj_[(size_t)i].id2), i});
    printf("%d out of %d pairs are redux pairs.\n", (int)redux_traj_.size(), (int)corres_traj_.size());
  }

  bool Blacklisted(const FramedTransformation& t) const { return blacklist_.count(t.id1) || blacklist_.count(t.id2); }

  // The reference's two loops are "#pragma omp parallel for" over the pair list (CorresApp.cpp:121,220).  Here each
  // GPU takes the pairs i with i % gpus == g (one host thread per GPU) and hands them to the *_batch entry points
  // in chunks; inside a chunk the library pipelines the pairs over several streams.
  static constexpr int kChunk = 64;

  bool Registration() {                                                // CorresApp.cpp:212-319
    registration_ = true;
    printf("Registration with dist %.6f, num %d and ratio %.6f\n", reg_dist_, reg_num_, reg_ratio_);
    int nprocessed = 0;
    bool ok = true;
#pragma omp parallel for num_threads(gpus_) schedule(static, 1) reduction(+ : nprocessed)
    for (int g = 0; g < gpus_; g++) {
      std::vector<int> live;
      for (int i = g; i < (int)corres_traj_.size(); i += gpus_) {
        FramedTransformation& ft = corres_traj_[(size_t)i];
        if (Blacklisted(ft)) {
          nprocessed++;
          ft.frame = -1;
          printf("Blacklist pair <%d, %d> ... \n", ft.id1, ft.id2);
          continue;
        }
        if (ft.frame == -1) {
          nprocessed++;
          continue;
        }
        live.push_back(i);
      }
      for (size_t c0 = 0; c0 < live.size(); c0 += kChunk) {
        const int m = (int)std::min<size_t>(kChunk, live.size() - c0);
        std::vector<er_cloud_t> src((size_t)m), tgt((size_t)m);
        std::vector<double> T((size_t)m * 16);
        std::vector<int> cnt((size_t)m, 0);
        for (int k = 0; k < m; k++) {
          const FramedTransformation& ft = corres_traj_[(size_t)live[c0 + (size_t)k]];
          tgt[(size_t)k] = pointclouds_[(size_t)g][(size_t)ft.id1];    // pcd0 = target, pcd1 = source (:238,:250)
          src[(size_t)k] = pointclouds_[(size_t)g][(size_t)ft.id2];
          memcpy(&T[(size_t)k * 16], ft.T, sizeof ft.T);
        }
        if (er_icp_count_inliers_batch(m, src.data(), tgt.data(), T.data(), reg_dist_, cnt.data()) != 0) {   // :249-264
          fprintf(stderr, "BuildCorrespondence: %s\n", er_last_error());
#pragma omp atomic write
          ok = false;
          continue;
        }
        std::vector<int> todo;
        for (int k = 0; k < m; k++) {
          FramedTransformation& ft = corres_traj_[(size_t)live[c0 + (size_t)k]];
          const int n0 = er_cloud_size(tgt[(size_t)k]), n1 = er_cloud_size(src[(size_t)k]);
          const double r1 = (double)cnt[(size_t)k] / (double)n0, r2 = (double)cnt[(size_t)k] / (double)n1;
          const bool accept = (cnt[(size_t)k] >= reg_num_ || (r1 > reg_ratio_ && r2 > reg_ratio_));   // :267
          printf("    <%d, %d> : %d inliers with ratio %.2f(%d) and %.2f(%d) ... %s\n", ft.id1, ft.id2, cnt[(size_t)k], r1, n0, r2, n1,
                 accept ? "accept." : "reject.");
          if (!accept) {
            ft.frame = -1;
            nprocessed++;
            continue;
          }
          ft.frame = cnt[(size_t)k];
          if (redux_) {                                                // :283-293
            auto it = redux_map_.find(GetReduxIndex(ft.id1, ft.id2));
            if (it != redux_map_.end()) {
              memcpy(ft.T, redux_traj_[(size_t)it->second].T, sizeof ft.T);
              nprocessed++;
              continue;
            }
          }
          todo.push_back(k);
        }
        const int a = (int)todo.size();
        if (a == 0) continue;
        std::vector<er_cloud_t> asrc((size_t)a), atgt((size_t)a);
        std::vector<float> guess((size_t)a * 16), fin((size_t)a * 16);
        std::vector<int> iters((size_t)a, 0), conv((size_t)a, 0);
        std::vector<double> fitness((size_t)a, 0.0);
        for (int q = 0; q < a; q++) {
          const int k = todo[(size_t)q];
          asrc[(size_t)q] = src[(size_t)k];
          atgt[(size_t)q] = tgt[(size_t)k];
          for (int e = 0; e < 16; e++) guess[(size_t)q * 16 + (size_t)e] = (float)T[(size_t)k * 16 + (size_t)e];   // transformation_.cast<float>(), :306
        }
        if (er_icp_align_batch(a, asrc.data(), atgt.data(), guess.data(), reg_dist_, 20, 1e-6, stop_rule_, fin.data(), iters.data(),
                               conv.data(), fitness.data()) != 0) {    // :295-306
          fprintf(stderr, "BuildCorrespondence: %s\n", er_last_error());
#pragma omp atomic write
          ok = false;
          continue;
        }
        for (int q = 0; q < a; q++) {
          FramedTransformation& ft = corres_traj_[(size_t)live[c0 + (size_t)todo[(size_t)q]]];
          printf("    <%d, %d> : ICP fitness score is %.6f (%d iterations)\n", ft.id1, ft.id2, fitness[(size_t)q], iters[(size_t)q]);
          for (int e = 0; e < 16; e++) ft.T[e] = (double)fin[(size_t)q * 16 + (size_t)e];   // getFinalTransformation().cast<double>(), :312
          nprocessed++;
        }
      }
    }
    printf("%d / %d\n", nprocessed, (int)corres_traj_.size());
    return ok;
  }

  bool FindCorrespondence() {                                          // CorresApp.cpp:112-210
    if (output_information_) {
      corres_info_.clear();
      for (const auto& t : corres_traj_) {
        FramedInformation fi;
        fi.id1 = t.id1; fi.id2 = t.id2; fi.frame = t.frame;
        memset(fi.info, 0, sizeof fi.info);
        corres_info_.push_back(fi);
      }
    }
    bool ok = true;
    omp_set_max_active_levels(2);                                      // the corres_*.txt writers below nest inside the per-GPU threads
#pragma omp parallel for num_threads(gpus_) schedule(static, 1)
    for (int g = 0; g < gpus_; g++) {
      std::vector<int> live;
      for (int i = g; i < (int)corres_traj_.size(); i += gpus_) {
        const FramedTransformation& ft = corres_traj_[(size_t)i];
        if (Blacklisted(ft) || ft.frame == -1) continue;
        live.push_back(i);
      }
      // The lists of a chunk land in ONE page-locked block (er_host_alloc), reused by the next chunk: the library writes them in place
      // (include/er_hip.h: no staging pass, no second copy), and pageable vectors would cost a page fault per 4 KB on top.
      int* block = nullptr;
      size_t block_ints = 0;
      for (size_t c0 = 0; c0 < live.size(); c0 += kChunk) {
        const int m = (int)std::min<size_t>(kChunk, live.size() - c0);
        std::vector<er_cloud_t> src((size_t)m), tgt((size_t)m);
        std::vector<double> T((size_t)m * 16), info((size_t)m * 36, 0.0);
        std::vector<int*> bufs((size_t)m);
        std::vector<int> cap((size_t)m), n((size_t)m, 0);
        std::vector<size_t> at((size_t)m);
        size_t need = 0;
        for (int k = 0; k < m; k++) {
          const FramedTransformation& ft = corres_traj_[(size_t)live[c0 + (size_t)k]];
          printf("Processing pair <%d, %d>\n", ft.id1, ft.id2);
          tgt[(size_t)k] = pointclouds_[(size_t)g][(size_t)ft.id1];
          src[(size_t)k] = pointclouds_[(size_t)g][(size_t)ft.id2];
          memcpy(&T[(size_t)k * 16], ft.T, sizeof ft.T);
          cap[(size_t)k] = std::max(er_cloud_size(src[(size_t)k]), 1);
          at[(size_t)k] = need;
          need += ((size_t)cap[(size_t)k] * 2 + 63) & ~(size_t)63;       // 256-byte aligned lists
        }
        if (need > block_ints) {
          if (block) er_host_free(block);
          block = (int*)er_host_alloc(need * sizeof(int));
          block_ints = block ? need : 0;
          if (!block) {
            fprintf(stderr, "BuildCorrespondence: %s\n", er_last_error());
#pragma omp atomic write
            ok = false;
            break;
          }
        }
        for (int k = 0; k < m; k++) bufs[(size_t)k] = block + at[(size_t)k];
        const auto tc0 = std::chrono::steady_clock::now();
        if (er_find_correspondence_batch(m, src.data(), tgt.data(), T.data(), dist_thresh_, normal_thresh_, bufs.data(), cap.data(), n.data(),
                                         output_information_ ? info.data() : nullptr) != 0) {
          fprintf(stderr, "BuildCorrespondence: %s\n", er_last_error());
#pragma omp atomic write
          ok = false;
          continue;
        }
        const auto tc1 = std::chrono::steady_clock::now();
#pragma omp parallel for num_threads(8) schedule(dynamic)
        for (int k = 0; k < m; k++) {
          const int i = live[c0 + (size_t)k];
          FramedTransformation& ft = corres_traj_[(size_t)i];
          const int nk = n[(size_t)k];
          printf("    <%d, %d> : Corresponce number is %d, ratio is %.2f(%d)\n", ft.id1, ft.id2, nk, (double)nk / (double)ft.frame, ft.frame);
          if ((double)nk / (double)ft.frame < 0.5) {                   // :164-171
            printf("    <%d, %d> : Reduced too much!!\n", ft.id1, ft.id2);
            ft.frame = reg_num_ > 0 ? -1 : nk;
          } else {
            ft.frame = nk;
          }
          if (save_corres_) {                                          // :175-184
            char fn[1024];
            snprintf(fn, sizeof fn, "%scorres_%d_%d.txt", m_pDirName.c_str(), ft.id1, ft.id2);
            if (!write_pair_lines(fn, bufs[(size_t)k], nk)) fprintf(stderr, "BuildCorrespondence: cannot write %s\n", fn);
          }
          if (output_information_) {                                   // :186-208
            corres_info_[(size_t)i].frame = ft.frame;
            memcpy(corres_info_[(size_t)i].info, &info[(size_t)k * 36], 36 * sizeof(double));
          }
        }
        if (getenv("ER_TIMING"))
          fprintf(stderr, "[timing]   chunk of %d pairs: er_find_correspondence_batch %.1f ms, corres_*.txt %.1f ms\n", m,
                  std::chrono::duration<double, std::milli>(tc1 - tc0).count(),
                  std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc1).count());
      }
      if (block) er_host_free(block);
    }
    return ok;
  }

  void Finalize() {                                                    // CorresApp.cpp:321-328
    erfmt::save_log("reg_output.log", corres_traj_);
    if (output_information_) erfmt::save_info("reg_output.info", corres_info_);
  }
};

}  // namespace

int main(int argc, char* argv[]) {
  erfmt::stage_done("process start");
  er_request_hw_queues(8);                                  // before the first HIP call (include/er_hip.h)
  using namespace erfmt;
  if (argc == 1 || find_switch(argc, argv, "--help") || find_switch(argc, argv, "-h")) return print_help();

  // The HIP runtime takes ~0.1 s to come up: it does so on a side thread while this one reads the fragment files.
  std::future<int> hip_up = std::async(std::launch::async, [] { return er_device_count(); });

  App app;
  if (find_switch(argc, argv, "--save_xyzn")) app.save_xyzn_ = true;
  std::string log_file, reg_log_file, blacklist_file, redux_file;
  double reg_dist, reg_ratio;
  int reg_num, num = 0;
  parse_argument(argc, argv, "--gpus", app.gpus_);
  parse_argument(argc, argv, "--device", app.device0_);
  parse_argument(argc, argv, "--icp_stop_rule", app.stop_rule_);
  if (app.gpus_ < 1) app.gpus_ = 1;

  int rc = 0;
  const bool job = (parse_argument(argc, argv, "--traj", log_file) > 0 && parse_argument(argc, argv, "--num", num) > 0) ||
                   parse_argument(argc, argv, "--reg_traj", reg_log_file) > 0;
  const auto t0 = std::chrono::steady_clock::now();
  bool ok = true;
  if (job) {
    if (parse_argument(argc, argv, "--reg_dist", reg_dist) > 0) {
      app.reg_dist_ = reg_dist;
      app.dist_thresh_ = reg_dist / 2.0;                               // BuildCorrespondence.cpp:52-55
    }
    if (parse_argument(argc, argv, "--reg_ratio", reg_ratio) > 0) app.reg_ratio_ = reg_ratio;
    if (parse_argument(argc, argv, "--reg_num", reg_num) > 0) app.reg_num_ = reg_num;
    parse_argument(argc, argv, "--length", app.length_);
    parse_argument(argc, argv, "--interval", app.interval_);
    ok = reg_log_file.length() > 0 ? app.LoadData(reg_log_file, -1) : app.LoadData(log_file, num);
    stage_done("LoadData (PCD files)");
  }
  const int visible = hip_up.get();
  stage_done("HIP runtime up (rest of the wait)");
  if (visible <= 0) {
    fprintf(stderr, "BuildCorrespondence: no HIP device available (there is no CPU fallback)\n");
    return 1;
  }
  if (app.device0_ + app.gpus_ > visible) app.gpus_ = std::max(1, visible - app.device0_);
  if (job) {
    ok = ok && app.Upload();
    stage_done("Upload (er_cloud_create_batch)");
    if (ok) {
      if (parse_argument(argc, argv, "--blacklist", blacklist_file) > 0 || parse_argument(argc, argv, "--blasklist", blacklist_file) > 0)
        app.Blacklist(blacklist_file);
      if (parse_argument(argc, argv, "--redux", redux_file) > 0) app.Redux(redux_file);
      if (find_switch(argc, argv, "--output_information")) app.output_information_ = true;
      if (find_switch(argc, argv, "--registration")) {
        const auto t1 = std::chrono::steady_clock::now();
        ok = app.Registration();
        std::cerr << "Neat Registration took " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count() << "ms." << std::endl;
      }
      stage_done("Registration");
      ok = app.FindCorrespondence() && ok;
      stage_done("FindCorrespondence + corres");
      app.Finalize();
      stage_done("Finalize (reg_output.*)");
    }
    std::cerr << "Registration All took " << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() << "ms." << std::endl;
    rc = ok ? 0 : 1;
  }
  // Every output file is closed: leave without the destructors (clouds, pooled workspaces, the HIP runtime's own teardown: ~0.1 s that a
  // pipeline script would wait for; the driver reclaims the device memory of a process that exits).
  fflush(nullptr);
  _exit(rc);
}
