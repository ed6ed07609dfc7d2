// FragmentOptimizer -- drop-in for the reference's FragmentOptimizer program (FragmentOptimizer/FragmentOptimizer.cpp,
// OptApp.{h,cpp}) on MI355X: same flags, same input files (rgbdslam / registration .log, cloud_bin_<i>.pcd or
// cloud_bin_xyzn_<i>.xyzn, corres_<i>_<j>.txt), same output files (--save_to .ctr, pose.log).
// The data-parallel half -- point set-up, UpdatePose / UpdateAllPointPN / UpdateAllNormal and the Hessian assembly of the
// three modes -- runs in liber_hip.so (er_fopt_*, csrc/er_fopt.hip).  The host keeps what the reference keeps on the host:
// the lattice regularizer, gauge terms, the pose / lattice updates and the linear solve.  The reference solves with CHOLMOD
// (sparse supernodal LL^T); here the SLAC and non-rigid systems are assembled, factored and solved in HBM by a dense Cholesky
// (er_fopt_factor_* / er_fopt_solve: the library's own blocked Cholesky over rocBLAS), the small rigid system (6 num unknowns) by a dense Cholesky on the host.  The non-rigid mode's system has num * 2187 unknowns: up to
// --dense_limit unknowns (default 30000) it is one dense matrix in HBM, beyond that the library keeps and factors it as the
// block-sparse lower triangle of fragment blocks (own potrf / rocBLAS trsm, syrk, gemm per 2187 x 2187 block; fill-in
// from a symbolic pass over the fragment graph) -- a 100-fragment scene needs a few hundred 38 MB blocks when its pairs link
// neighbours, 193 GB for the complete graph, both inside one MI355X.  A failed allocation is reported with a clear message.
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <set>
#include <string>
#include <vector>

#include "../../../include/er_hip.h"
#include "er_formats.h"

namespace {

using erfmt::FramedTransformation;

typedef std::vector<double> Vec;

void mat4_mul(const double* A, const double* B, double* C) {
  double t[16];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) t[r * 4 + c] = ((A[r * 4] * B[c] + A[r * 4 + 1] * B[4 + c]) + A[r * 4 + 2] * B[8 + c]) + A[r * 4 + 3] * B[12 + c];
  memcpy(C, t, sizeof t);
}

bool mat4_inverse(const double* m, double* inv) {      // cofactor expansion
  double a[16];
  a[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  a[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  a[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  a[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  a[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  a[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  a[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  a[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  a[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  a[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  a[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  a[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  a[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  a[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  a[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  a[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const double det = m[0] * a[0] + m[1] * a[4] + m[2] * a[8] + m[3] * a[12];
  if (det == 0.0 || !std::isfinite(det)) return false;
  for (int i = 0; i < 16; i++) inv[i] = a[i] / det;
  return true;
}

// Symmetric dense solve A x = b by Cholesky (A: n x n row-major, only the LOWER triangle is read; destroyed).  Returns
// false when A is not positive definite.
bool cholesky_solve(std::vector<double>& A, long n, Vec& b) {
  for (long k = 0; k < n; k++) {
    const double piv = A[(size_t)k * n + k];
    if (!(piv > 0.0)) return false;
    const double s = std::sqrt(piv);
    A[(size_t)k * n + k] = s;
    for (long i = k + 1; i < n; i++) A[(size_t)i * n + k] /= s;
#pragma omp parallel for schedule(dynamic, 16) num_threads(8) if (n - k > 256)
    for (long i = k + 1; i < n; i++) {
      const double lik = A[(size_t)i * n + k];
      if (lik == 0.0) continue;
      double* Ai = &A[(size_t)i * n];
      for (long j = k + 1; j <= i; j++) Ai[j] -= lik * A[(size_t)j * n + k];
    }
  }
  for (long i = 0; i < n; i++) {
    double s = b[(size_t)i];
    for (long j = 0; j < i; j++) s -= A[(size_t)i * n + j] * b[(size_t)j];
    b[(size_t)i] = s / A[(size_t)i * n + i];
  }
  for (long i = n - 1; i >= 0; i--) {
    double s = b[(size_t)i];
    for (long j = i + 1; j < n; j++) s -= A[(size_t)j * n + i] * b[(size_t)j];
    b[(size_t)i] = s / A[(size_t)i * n + i];
  }
  return true;
}

// COptApp::GetRotation, OptApp.cpp:850-871: C = sum dif^T diff, R = V U^T of its SVD (last column of U flipped when det < 0)
// = the proper rotation maximising trace(R C).  Computed from the symmetric 4x4 matrix of Horn's closed form with Jacobi
// eigen-iterations (no SVD library here); equal to the SVD construction wherever that one is well defined.
void best_rotation(const double C[9], double R[9]) {
  // R maximises sum_i diff_i . (R dif_i)  with C = sum dif_i^T diff_i  (C[a][b] = sum dif[a] * diff[b])
  const double Sxx = C[0], Sxy = C[1], Sxz = C[2], Syx = C[3], Syy = C[4], Syz = C[5], Szx = C[6], Szy = C[7], Szz = C[8];
  double N[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                    {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                    {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                    {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = 0.0;
    for (int p = 0; p < 4; p++)
      for (int q = p + 1; q < 4; q++) off += N[p][q] * N[p][q];
    if (off < 1e-300) break;
    for (int p = 0; p < 4; p++)
      for (int q = p + 1; q < 4; q++) {
        if (N[p][q] == 0.0) continue;
        const double theta = (N[q][q] - N[p][p]) / (2.0 * N[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; k++) {
          const double a = N[k][p], b = N[k][q];
          N[k][p] = c * a - s * b;
          N[k][q] = s * a + c * b;
        }
        for (int k = 0; k < 4; k++) {
          const double a = N[p][k], b = N[q][k];
          N[p][k] = c * a - s * b;
          N[q][k] = s * a + c * b;
        }
        for (int k = 0; k < 4; k++) {
          const double a = V[k][p], b = V[k][q];
          V[k][p] = c * a - s * b;
          V[k][q] = s * a + c * b;
        }
      }
  }
  int best = 0;
  for (int k = 1; k < 4; k++)
    if (N[k][k] > N[best][best]) best = k;
  double q0 = V[0][best], qx = V[1][best], qy = V[2][best], qz = V[3][best];
  const double nn = std::sqrt(q0 * q0 + qx * qx + qy * qy + qz * qz);
  q0 /= nn; qx /= nn; qy /= nn; qz /= nn;
  R[0] = q0 * q0 + qx * qx - qy * qy - qz * qz; R[1] = 2 * (qx * qy - q0 * qz); R[2] = 2 * (qx * qz + q0 * qy);
  R[3] = 2 * (qy * qx + q0 * qz); R[4] = q0 * q0 - qx * qx + qy * qy - qz * qz; R[5] = 2 * (qy * qz - q0 * qx);
  R[6] = 2 * (qz * qx - q0 * qy); R[7] = 2 * (qz * qy + q0 * qx); R[8] = q0 * q0 - qx * qx - qy * qy + qz * qz;
}

class COptApp {                                     // OptApp.h:39-124
 public:
  std::vector<FramedTransformation> rgbd_traj_, reg_traj_;
  int resolution_ = 8, interval_ = 50, num_ = 0;
  double weight_ = 1.0, length_ = 3.0;
  int max_iteration_ = 5, max_inner_iteration_ = 10;
  std::string dir_prefix_, ctr_filename_ = "output.ctr", pose_filename_ = "pose.log", init_ctr_file_, sample_filename_ = "sample.pcd";
  int sample_num_ = -1, blacklist_pair_num_ = 10000, device_ = 0;
  long dense_limit_ = 30000;
  std::set<int> blacklist_;
  std::vector<int> absolute2relative_map_, relative2absolute_map_;
  std::vector<std::vector<double>> ipose_, pose_;    // 16 doubles each, row-major
  er_fopt_t fo_ = nullptr;
  int nper_ = 0, nv_ = 0;
  double unit_length_ = 0;
  struct Edge { int v; std::vector<int> nb; int i, j, k; };
  std::vector<Edge> edges_;

  ~COptApp() { if (fo_) er_fopt_destroy(fo_); }

  int GetIndex(int i, int j, int k) const { return i + j * (resolution_ + 1) + k * (resolution_ + 1) * (resolution_ + 1); }

  void Blacklist(const std::string& fn) {           // OptApp.cpp:925-941
    blacklist_.clear();
    if (FILE* f = fopen(fn.c_str(), "r")) {
      char buf[1024];
      int id;
      while (fgets(buf, 1024, f))
        if (strlen(buf) > 0 && buf[0] != '#' && sscanf(buf, "%d", &id) == 1) blacklist_.insert(id);
      fclose(f);
    }
  }

  void IPoseFromFile(const std::string& fn) {       // OptApp.h:91-98
    std::vector<FramedTransformation> ip;
    erfmt::load_log(fn, ip);
    ipose_.clear();
    for (const auto& t : ip) ipose_.emplace_back(t.T, t.T + 16);
  }

  void InitMap() {                                  // OptApp.cpp:31-47
    absolute2relative_map_.assign((size_t)num_, -1);
    relative2absolute_map_.clear();
    for (int i = 0; i < num_; i++)
      if (!blacklist_.count(i)) {
        absolute2relative_map_[(size_t)i] = (int)relative2absolute_map_.size();
        relative2absolute_map_.push_back(i);
      }
    if (num_ != (int)relative2absolute_map_.size())
      printf("Blacklisted fragments, number reduced from %d to %d.\n", num_, (int)relative2absolute_map_.size());
    num_ = (int)relative2absolute_map_.size();
  }

  bool InitIPose() {                                // OptApp.cpp:49-72 (pose_ is sized in every case; the reference forgets to with --ipose)
    pose_.assign((size_t)num_, std::vector<double>(16, 0.0));
    if (!ipose_.empty()) {
      if ((int)ipose_.size() < num_) { fprintf(stderr, "FragmentOptimizer: --ipose holds %zu poses, %d needed\n", ipose_.size(), num_); return false; }
      return true;
    }
    double base[16] = {1, 0, 0, length_ / 2.0, 0, 1, 0, length_ / 2.0, 0, 0, 1, -0.3, 0, 0, 0, 1}, binv[16], r0inv[16], left[16];
    if (rgbd_traj_.empty()) { fprintf(stderr, "FragmentOptimizer: no --rgbdslam / --ipose trajectory\n"); return false; }
    mat4_inverse(base, binv);
    if (!mat4_inverse(rgbd_traj_[0].T, r0inv)) return false;
    mat4_mul(base, r0inv, left);
    ipose_.assign((size_t)num_, std::vector<double>(16));
    for (int i = 0; i < num_; i++) {
      const size_t k = (size_t)relative2absolute_map_[(size_t)i] * (size_t)interval_;
      if (k >= rgbd_traj_.size()) { fprintf(stderr, "FragmentOptimizer: trajectory too short for fragment %d\n", i); return false; }
      double t[16];
      mat4_mul(left, rgbd_traj_[k].T, t);
      mat4_mul(t, binv, ipose_[(size_t)i].data());
    }
    printf("IPose initialized.\n");
    return true;
  }

  bool InitPointClouds() {                          // OptApp.cpp:74-98
    if (er_fopt_create(num_, resolution_, (float)length_, device_, &fo_)) { fprintf(stderr, "FragmentOptimizer: %s\n", er_last_error()); return false; }
    for (int i = 0; i < num_; i++) {
      const int ii = relative2absolute_map_[(size_t)i];
      char fn[1024];
      std::vector<float> xyz, nrm;
      snprintf(fn, sizeof fn, "%scloud_bin_%d.pcd", dir_prefix_.c_str(), ii);
      if (erfmt::file_exists(fn)) {
        std::vector<std::vector<float>> cols;
        size_t n = 0;
        if (!erfmt::load_pcd_fields(fn, {"x", "y", "z", "normal_x", "normal_y", "normal_z"}, cols, n)) { fprintf(stderr, "Error loading file.\n"); return false; }
        for (size_t k = 0; k < n; k++)
          if (!std::isnan(cols[3][k])) {              // PointCloud.cpp:29
            xyz.insert(xyz.end(), {cols[0][k], cols[1][k], cols[2][k]});
            nrm.insert(nrm.end(), {cols[3][k], cols[4][k], cols[5][k]});
          }
      } else {
        snprintf(fn, sizeof fn, "%scloud_bin_xyzn_%d.xyzn", dir_prefix_.c_str(), ii);
        FILE* f = fopen(fn, "r");
        if (!f) { fprintf(stderr, "File not found ... Check dir and num parameters.\n"); return false; }
        char buf[1024];
        float x[6] = {0, 0, 0, 0, 0, 0};
        while (fgets(buf, 1024, f))
          if (strlen(buf) > 0 && buf[0] != '#') {     // PointCloud.cpp:49-52
            sscanf(buf, "%f %f %f %f %f %f", &x[0], &x[1], &x[2], &x[3], &x[4], &x[5]);
            xyz.insert(xyz.end(), {x[0], x[1], x[2]});
            nrm.insert(nrm.end(), {x[3], x[4], x[5]});
          }
        fclose(f);
      }
      int bad = -1;
      if (er_fopt_set_cloud(fo_, i, xyz.data(), nrm.data(), (int)(xyz.size() / 3), &bad)) { fprintf(stderr, "FragmentOptimizer: %s\n", er_last_error()); return false; }
      if (bad >= 0) fprintf(stderr, "Error!! Point out of bound!!\n");
      printf("Read %s ... get %d points.\n", fn, er_fopt_cloud_size(fo_, i));
    }
    return true;
  }

  bool InitCorrespondences() {                      // OptApp.cpp:100-118
    std::vector<int> fi, fj, counts;
    std::vector<std::vector<int>> rows;
    for (const auto& t : reg_traj_) {
      // OptApp.cpp:104: the ids are compared with num_ AFTER InitMap reduced it to the non-blacklisted count (so with a
      // blacklist the highest absolute ids are dropped too) -- mirrored, the output files must equal the reference program's
      if (t.id1 < 0 || t.id2 < 0 || blacklist_.count(t.id1) || blacklist_.count(t.id2) || t.id1 >= num_ || t.id2 >= num_ ||
          t.id1 >= (int)absolute2relative_map_.size() || t.id2 >= (int)absolute2relative_map_.size())
        continue;
      if (t.frame != -1 && t.frame >= blacklist_pair_num_) {
        char fn[1024];
        snprintf(fn, sizeof fn, "%scorres_%d_%d.txt", dir_prefix_.c_str(), t.id1, t.id2);
        std::vector<int> r;
        if (FILE* f = fopen(fn, "r")) {
          char buf[1024];
          int a, b;
          while (fgets(buf, 1024, f))
            if (strlen(buf) > 0 && buf[0] != '#' && sscanf(buf, "%d %d", &a, &b) == 2) { r.push_back(a); r.push_back(b); }
          fclose(f);
        }
        printf("Read %s, get %d correspondences.\n", fn, (int)r.size() / 2);
        fi.push_back(absolute2relative_map_[(size_t)t.id1]);
        fj.push_back(absolute2relative_map_[(size_t)t.id2]);
        counts.push_back((int)r.size() / 2);
        rows.push_back(std::move(r));
      }
    }
    std::vector<const int*> ptrs;
    for (auto& r : rows) ptrs.push_back(r.data());
    if (er_fopt_set_correspondences(fo_, (int)fi.size(), fi.data(), fj.data(), ptrs.data(), counts.data())) {
      fprintf(stderr, "FragmentOptimizer: %s\n", er_last_error());
      return false;
    }
    return true;
  }

  void InitLattice() {
    unit_length_ = length_ / resolution_;
    nv_ = (resolution_ + 1) * (resolution_ + 1) * (resolution_ + 1);
    nper_ = nv_ * 3;
    edges_.clear();
    const int r = resolution_;
    for (int i = 0; i <= r; i++)
      for (int j = 0; j <= r; j++)
        for (int k = 0; k <= r; k++) {                // the reference's six `if` blocks, in order
          Edge e;
          e.v = GetIndex(i, j, k); e.i = i; e.j = j; e.k = k;
          if (i > 0) e.nb.push_back(GetIndex(i - 1, j, k));
          if (i < r) e.nb.push_back(GetIndex(i + 1, j, k));
          if (j > 0) e.nb.push_back(GetIndex(i, j - 1, k));
          if (j < r) e.nb.push_back(GetIndex(i, j + 1, k));
          if (k > 0) e.nb.push_back(GetIndex(i, j, k - 1));
          if (k < r) e.nb.push_back(GetIndex(i, j, k + 1));
          edges_.push_back(e);
        }
  }

  // AddHessian2( {v, nb}, {1, -1} ) for every (vertex, neighbour): [[1,-1],[-1,1]] per component, added at offset `off`
  void add_laplacian(std::vector<double>& A, long n, long off, double scale) const {
    for (const Edge& e : edges_)
      for (int w : e.nb)
        for (int c = 0; c < 3; c++) {
          const long a = off + e.v * 3 + c, b = off + w * 3 + c;
          A[(size_t)a * n + a] += scale;
          A[(size_t)b * n + b] += scale;
          A[(size_t)a * n + b] -= scale;
          A[(size_t)b * n + a] -= scale;
        }
  }

  void rotation_of(const double* ini, const double* cur, const Edge& e, double R[9]) const {
    double C[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int w : e.nb)
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) C[a * 3 + b] += (ini[e.v * 3 + a] - ini[w * 3 + a]) * (cur[e.v * 3 + b] - cur[w * 3 + b]);
    best_rotation(C, R);
  }

  static void increment(const double* x6, double* aff) {   // AngleAxis(z) * AngleAxis(y) * AngleAxis(x), OptApp.cpp:395-400
    const double a = x6[0], b = x6[1], g = x6[2];
    const double Rx[9] = {1, 0, 0, 0, cos(a), -sin(a), 0, sin(a), cos(a)}, Ry[9] = {cos(b), 0, sin(b), 0, 1, 0, -sin(b), 0, cos(b)},
                 Rz[9] = {cos(g), -sin(g), 0, sin(g), cos(g), 0, 0, 0, 1};
    double zy[9], R[9];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) zy[r * 3 + c] = (Rz[r * 3] * Ry[c] + Rz[r * 3 + 1] * Ry[3 + c]) + Rz[r * 3 + 2] * Ry[6 + c];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) R[r * 3 + c] = (zy[r * 3] * Rx[c] + zy[r * 3 + 1] * Rx[3 + c]) + zy[r * 3 + 2] * Rx[6 + c];
    for (int i = 0; i < 16; i++) aff[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) aff[r * 4 + c] = R[r * 3 + c];
      aff[r * 4 + 3] = x6[3 + r];
    }
  }

  bool update_pose_gpu(int l, const double* M) {
    float Mf[16];
    for (int i = 0; i < 16; i++) Mf[i] = (float)M[i];
    if (er_fopt_update_pose(fo_, l, Mf)) { fprintf(stderr, "FragmentOptimizer: %s\n", er_last_error()); return false; }
    return true;
  }

  void canonical_lattice(Vec& out) const {           // InitCtrSLAC, OptApp.cpp:723-733
    out.assign((size_t)nper_, 0.0);
    for (const Edge& e : edges_) {
      out[(size_t)e.v * 3] = e.i * unit_length_;
      out[(size_t)e.v * 3 + 1] = e.j * unit_length_;
      out[(size_t)e.v * 3 + 2] = e.k * unit_length_;
    }
  }

  static void apply(const double* P, const double* xyz, double* out) {
    for (int r = 0; r < 3; r++) out[r] = ((P[r * 4] * xyz[0] + P[r * 4 + 1] * xyz[1]) + P[r * 4 + 2] * xyz[2]) + P[r * 4 + 3] * 1.0;
  }

  void expand(const Vec& lattice, Vec& out) const {  // Pose2Ctr / ExpandCtr, OptApp.cpp:735-763
    out.assign((size_t)num_ * nper_, 0.0);
    for (int l = 0; l < num_; l++)
      for (int v = 0; v < nv_; v++) apply(pose_[(size_t)l].data(), &lattice[(size_t)v * 3], &out[(size_t)l * nper_ + (size_t)v * 3]);
  }

  // --init_ctr: `count` lines "%lf %lf %lf" into out[offset ...] (InitCtr :693-707, InitCtrSLAC :734-745)
  bool LoadInitCtr(Vec& out, size_t offset, size_t count) const {
    FILE* f = fopen(init_ctr_file_.c_str(), "r");
    if (!f) return false;                               // the reference silently keeps its default too
    char buf[1024];
    for (size_t i = 0; i < count; i++) {
      if (!fgets(buf, 1024, f)) break;
      sscanf(buf, "%lf %lf %lf", &out[offset + i * 3], &out[offset + i * 3 + 1], &out[offset + i * 3 + 2]);
    }
    fclose(f);
    return true;
  }

  void SaveCtr(const Vec& ctr, const std::string& fn) const {   // OptApp.cpp:873-884
    printf("Save ctr to file %s ... ", fn.c_str());
    if (FILE* f = fopen(fn.c_str(), "w")) {
      for (size_t i = 0; i < ctr.size() / 3; i++) fprintf(f, "%.10f %.10f %.10f\n", ctr[i * 3], ctr[i * 3 + 1], ctr[i * 3 + 2]);
      fclose(f);
    }
    printf("Done.\n");
  }

  // SavePoints, OptApp.cpp:897-923: every sample_num_-th point of every fragment, moved by its lattice (UpdateAllNormal +
  // UpdatePoint = the device's UpdateAllPointPN), normal re-normalised in float64, written as sample.pcd.
  bool SavePoints(const Vec& ctr) {
    if (sample_num_ <= 0) return true;
    std::vector<float> col[8];
    std::vector<float> p, nrm;
    for (int l = 0; l < num_; l++) {
      if (er_fopt_update_point_pn(fo_, l, &ctr[(size_t)l * nper_])) { fprintf(stderr, "FragmentOptimizer: %s\n", er_last_error()); return false; }
      const int n = er_fopt_cloud_size(fo_, l);
      p.resize((size_t)n * 3);
      nrm.resize((size_t)n * 3);
      if (n && er_fopt_get_points(fo_, l, nullptr, nullptr, nullptr, p.data(), nrm.data())) { fprintf(stderr, "FragmentOptimizer: %s\n", er_last_error()); return false; }
      for (int i = 0; i < n; i += sample_num_) {
        double v[3] = {nrm[(size_t)i * 3], nrm[(size_t)i * 3 + 1], nrm[(size_t)i * 3 + 2]};
        const double inv = 1.0 / std::sqrt(v[0] * v[0] + (v[1] * v[1] + v[2] * v[2]));   // Eigen's normalize(): v *= 1 / norm()
        for (int c = 0; c < 3; c++) {
          col[c].push_back(p[(size_t)i * 3 + c]);
          col[3 + c].push_back((float)(v[c] * inv));
        }
        col[6].push_back(0.0f);
        col[7].push_back(0.0f);
      }
    }
    printf("Save sample pcd into %s ... ", sample_filename_.c_str());
    std::vector<const float*> cols;
    for (auto& c : col) cols.push_back(c.data());
    const bool ok = erfmt::save_pcd_compressed(sample_filename_, {"x", "y", "z", "normal_x", "normal_y", "normal_z", "rgb", "curvature"}, cols, col[0].size());
    printf("Done.\n");
    return ok;
  }

  void SavePoses() const {
    std::vector<FramedTransformation> out;
    for (int i = 0; i < num_; i++) {
      FramedTransformation t;
      t.id1 = i; t.id2 = i; t.frame = i + 1;
      memcpy(t.T, pose_[(size_t)i].data(), sizeof t.T);
      out.push_back(t);
    }
    erfmt::save_log(pose_filename_, out);
  }

  bool Prepare() {
    InitMap();
    InitLattice();
    return InitIPose() && InitPointClouds() && InitCorrespondences();
  }

  // ---- OptimizeRigid, OptApp.cpp:282-417 ------------------------------------------------------------------
  bool OptimizeRigid() {
    printf("Rigid optimization.\nParameters: resolution %d, piece number %d, max iteration %d\n", resolution_, num_, max_iteration_);
    if (!Prepare()) return false;
    const long N = 6L * num_;
    for (int i = 0; i < num_; i++) {
      pose_[(size_t)i] = ipose_[(size_t)i];
      if (!update_pose_gpu(i, pose_[(size_t)i].data())) return false;
    }
    std::vector<double> JJ((size_t)N * N);
    Vec Jb((size_t)N);
    for (int itr = 0; itr < max_iteration_; itr++) {
      double score = 0;
      if (er_fopt_assemble_rigid(fo_, JJ.data(), Jb.data(), &score)) { fprintf(stderr, "FragmentOptimizer: %s\n", er_last_error()); return false; }
      printf("Error score is : %.2f\n", score);
      if (!cholesky_solve(JJ, N, Jb)) { fprintf(stderr, "FragmentOptimizer: the rigid system is not positive definite\n"); return false; }
      for (int l = 0; l < num_; l++) {
        double x6[6], aff[16], np[16];
        for (int q = 0; q < 6; q++) x6[q] = -Jb[(size_t)l * 6 + q];           // result = - solver.solve( thisJb )
        increment(x6, aff);
        mat4_mul(aff, pose_[(size_t)l].data(), np);
        pose_[(size_t)l].assign(np, np + 16);
        if (!update_pose_gpu(l, aff)) return false;
      }
    }
    SavePoses();
    Vec lat, ctr;
    canonical_lattice(lat);
    expand(lat, ctr);
    SaveCtr(ctr, ctr_filename_);
    return SavePoints(ctr);
  }

  // ---- OptimizeSLAC, OptApp.cpp:419-680 ---------------------------------------------------------------------
  bool OptimizeSLAC() {
    printf("SLAC optimization.\nParameters: weight %.5f, resolution %d, piece number %d, max iteration %d\n", weight_, resolution_, num_, max_iteration_);
    const double default_weight = num_ * weight_;      // note: num_ BEFORE InitMap, like the reference (:421 precedes :425)
    if (!Prepare()) return false;
    const long N = 6L * num_ + nper_, L0 = 6L * num_;
    Vec ictr, thisCtr, expand_ctr;
    canonical_lattice(ictr);
    thisCtr = ictr;
    if (init_ctr_file_.length() > 1) LoadInitCtr(thisCtr, 0, (size_t)nv_);
    for (int i = 0; i < num_; i++) {
      pose_[(size_t)i] = ipose_[(size_t)i];
      if (!update_pose_gpu(i, pose_[(size_t)i].data())) return false;
    }
    Vec Jb((size_t)N), rot((size_t)num_ * 9);
    for (int itr = 0; itr < max_iteration_; itr++) {
      for (int l = 0; l < num_; l++)
        for (int r = 0; r < 3; r++)
          for (int c = 0; c < 3; c++) rot[(size_t)l * 9 + r * 3 + c] = pose_[(size_t)l][(size_t)c * 4 + r];     // pose_rot_t_ = R^T
      double score = 0;
      // thisJJ = Upper( baseJJ * default_weight ) + gauge + data term: assembled AND factored in HBM (the library's own Cholesky)
      if (er_fopt_factor_slac(fo_, rot.data(), default_weight, Jb.data(), &score)) { fprintf(stderr, "FragmentOptimizer: %s\n", er_last_error()); return false; }
      printf("Data error score is : %.2f\n", score);
      // regularizer right-hand side, OptApp.cpp:570-631
      Vec b(Jb);
      double regscore = 0;
      for (const Edge& e : edges_) {
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (!(e.i == resolution_ / 2 && e.j == resolution_ / 2 && e.k == 0)) rotation_of(ictr.data(), thisCtr.data(), e, R);
        for (int w : e.nb) {
          double bx[3];
          for (int a = 0; a < 3; a++) {
            double rd = 0;
            for (int c = 0; c < 3; c++) rd += R[a * 3 + c] * (ictr[(size_t)e.v * 3 + c] - ictr[(size_t)w * 3 + c]);
            bx[a] = (thisCtr[(size_t)e.v * 3 + a] - thisCtr[(size_t)w * 3 + a]) - rd;
            regscore += default_weight * bx[a] * bx[a];
            b[(size_t)(L0 + e.v * 3 + a)] += bx[a] * default_weight;
            b[(size_t)(L0 + w * 3 + a)] -= bx[a] * default_weight;
          }
        }
      }
      printf("Regularization error score is : %.2f\n", regscore);
      {
        Vec x((size_t)N);
        if (er_fopt_solve(fo_, b.data(), 0, x.data())) { fprintf(stderr, "FragmentOptimizer: %s\n", er_last_error()); return false; }
        b.swap(x);
      }
      for (int q = 0; q < nper_; q++) thisCtr[(size_t)q] += -b[(size_t)(L0 + q)];
      for (int l = 0; l < num_; l++) {
        double x6[6], aff[16], np[16];
        for (int q = 0; q < 6; q++) x6[q] = -b[(size_t)l * 6 + q];
        increment(x6, aff);
        mat4_mul(aff, pose_[(size_t)l].data(), np);
        pose_[(size_t)l].assign(np, np + 16);
      }
      expand(thisCtr, expand_ctr);
      for (int l = 0; l < num_; l++)
        if (er_fopt_update_point_pn(fo_, l, &expand_ctr[(size_t)l * nper_])) { fprintf(stderr, "FragmentOptimizer: %s\n", er_last_error()); return false; }
    }
    SavePoses();
    expand(thisCtr, expand_ctr);
    SaveCtr(expand_ctr, ctr_filename_);
    return SavePoints(expand_ctr);
  }

  // ---- OptimizeNonrigid, OptApp.cpp:120-278 -------------------------------------------------------------------
  bool OptimizeNonrigid() {
    printf("Nonrigid optimization.\nParameters: weight %.5f, resolution %d, piece number %d, max iteration %d\n", weight_, resolution_, num_, max_iteration_);
    if (!Prepare()) return false;
    const long M = (long)num_ * nper_;
    {
      char lim[32];
      snprintf(lim, sizeof lim, "%ld", dense_limit_);
      setenv("ER_FOPT_DENSE_MAX", lim, 1);                   // the library's switch between the dense and the block-sparse factorisation
      if (M > dense_limit_) printf("Non-rigid system of %ld unknowns: block-sparse Cholesky over %d fragment blocks.\n", M, num_);
    }
    Vec lat, ctr, ictr, oldctr;
    canonical_lattice(lat);
    for (int i = 0; i < num_; i++) pose_[(size_t)i] = ipose_[(size_t)i];
    expand(lat, ctr);                                 // InitCtr, :709-721
    if (init_ctr_file_.length() > 1) LoadInitCtr(ctr, 0, (size_t)num_ * nv_);
    ictr = ctr;
    oldctr = ctr;                                     // :143
    for (int itr = 0; itr < max_iteration_; itr++) {
      for (int l = 0; l < num_; l++)
        if (er_fopt_update_normals(fo_, l, &ctr[(size_t)l * nper_])) { fprintf(stderr, "FragmentOptimizer: %s\n", er_last_error()); return false; }
      // thisAA = baseAA + data blocks: scattered into a dense matrix and factored in HBM (the library's own Cholesky)
      if (er_fopt_factor_nonrigid(fo_, weight_)) { fprintf(stderr, "FragmentOptimizer: %s\n", er_last_error()); return false; }
      if (sample_num_ > 0) SaveCtr(oldctr, "itr" + std::to_string(itr) + ".ctr");           // :213-218 (oldctr: the lattice before the last inner solve)
      for (int m = 0; m < max_inner_iteration_; m++) {
        Vec Ab((size_t)M, 0.0);
        for (int l = 0; l < num_; l++) {
          const double* ini = &ictr[(size_t)l * nper_];
          const double* cur = &ctr[(size_t)l * nper_];
          for (const Edge& e : edges_) {              // :221-260
            double R[9];
            rotation_of(ini, cur, e, R);
            for (int w : e.nb)
              for (int a = 0; a < 3; a++) {
                double bx = 0;
                for (int c = 0; c < 3; c++) bx += R[a * 3 + c] * (ini[e.v * 3 + c] - ini[w * 3 + c]);
                Ab[(size_t)l * nper_ + (size_t)e.v * 3 + a] += bx;
                Ab[(size_t)l * nper_ + (size_t)w * 3 + a] -= bx;
              }
          }
        }
        {
          Vec x((size_t)M);
          if (er_fopt_solve(fo_, Ab.data(), 0, x.data())) { fprintf(stderr, "FragmentOptimizer: %s\n", er_last_error()); return false; }
          Ab.swap(x);
        }
        double sc = 0;
        for (long q = 0; q < M; q++) sc += (ctr[(size_t)q] - Ab[(size_t)q]) * (ctr[(size_t)q] - Ab[(size_t)q]);
        oldctr = ctr;                                 // :261
        ctr = Ab;
        printf("Iteration #%d:%d (%d:%d) : score is %.4f\n", itr + 1, m + 1, max_iteration_, max_inner_iteration_, std::sqrt(sc));
        if (sample_num_ > 0) SaveCtr(ctr, "itr" + std::to_string(itr) + "_inner" + std::to_string(m) + "_out.ctr");   // :267-272
      }
    }
    SaveCtr(ctr, ctr_filename_);
    return SavePoints(ctr);
  }
};

int print_help() {
  printf("\nApplication parameters:\n"
         "    --help, -h                      : print this message\n"
         "    --rgbdslam <log_file>           : rgbdslam.log/opt_output.log file, get ipose\n"
         "    --registration <log_file>       : reg_output.log, invalid pair when frame_ == -1\n"
         "    --dir <dir_prefix>              : dir prefix, place to loopup .xyzn files\n"
         "    --num <number>                  : number of pieces, important parameter\n"
         "    --weight <weight>               : 1.0 for nonrigid, 10000.0 for rigid\n"
         "    --resolution <resolution>       : default - 8\n"
         "    --length <length>               : default - 3.0\n"
         "    --interval <interval>           : default - 50\n"
         "    --iteration <max_number>        : default - 5\n"
         "    --inner_iteration <max_number>  : default - 10\n"
         "    --save_to <ctr_file>            : default - output.ctr\n"
         "    --init_ctr <ctr_file>           : initial control lattice(s)\n"
         "    --blasklist <blacklist_file>    : each line is the block we want to blacklist\n"
         "    --blacklistpair <threshold>     : threshold of accepting pairwise registration, default - 10000\n"
         "    --ipose <log_file>              : get ipose from log file\n"
         "    --write_xyzn_sample <sample_num>: per <sample_num> write a point into sample.pcd\n"
         "    --device <id>, --dense_limit <n>: (new) HIP device; largest non-rigid system factored as ONE dense matrix (beyond: block-sparse)\n"
         "Optimization options:\n"
         "    --nonrigid                      : default, nonrigid alignment published in ICCV 2013\n"
         "    --rigid                         : dense rigid optimization\n"
         "    --slac                          : simultaneous localization and calibration, published in CVPR 2014\n");
  return 0;
}

}  // namespace

int main(int argc, char* argv[]) {
  er_request_hw_queues(8);                                  // before the first HIP call (include/er_hip.h)                     // FragmentOptimizer.cpp:40-93
  using namespace erfmt;
  if (argc == 1 || find_switch(argc, argv, "--help") || find_switch(argc, argv, "-h")) return print_help();
  COptApp app;
  std::string s, blacklist_file, ipose_file;
  if (parse_argument(argc, argv, "--rgbdslam", s) > 0) load_log(s, app.rgbd_traj_);
  if (parse_argument(argc, argv, "--registration", s) > 0) load_log(s, app.reg_traj_);
  parse_argument(argc, argv, "--dir", app.dir_prefix_);
  parse_argument(argc, argv, "--init_ctr", app.init_ctr_file_);
  parse_argument(argc, argv, "--num", app.num_);
  parse_argument(argc, argv, "--weight", app.weight_);
  parse_argument(argc, argv, "--resolution", app.resolution_);
  parse_argument(argc, argv, "--length", app.length_);
  parse_argument(argc, argv, "--interval", app.interval_);
  parse_argument(argc, argv, "--iteration", app.max_iteration_);
  parse_argument(argc, argv, "--inner_iteration", app.max_inner_iteration_);
  parse_argument(argc, argv, "--save_to", app.ctr_filename_);
  parse_argument(argc, argv, "--write_xyzn_sample", app.sample_num_);
  parse_argument(argc, argv, "--blacklistpair", app.blacklist_pair_num_);
  parse_argument(argc, argv, "--device", app.device_);
  double dl = (double)app.dense_limit_;
  if (parse_argument(argc, argv, "--dense_limit", dl) > 0) app.dense_limit_ = (long)dl;
  if (parse_argument(argc, argv, "--blacklist", blacklist_file) > 0) app.Blacklist(blacklist_file);
  if (parse_argument(argc, argv, "--ipose", ipose_file) > 0) app.IPoseFromFile(ipose_file);
  bool ok;
  if (find_switch(argc, argv, "--slac")) ok = app.OptimizeSLAC();
  else if (find_switch(argc, argv, "--rigid")) ok = app.OptimizeRigid();
  else ok = app.OptimizeNonrigid();
  return ok ? 0 : 1;
}
