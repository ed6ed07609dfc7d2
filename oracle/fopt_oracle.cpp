// fopt_oracle.cpp -- CPU restatement of the data-parallel half of the reference's FragmentOptimizer
// (SURVEY.md 8f-2): point set-up, pose / control-lattice updates and the Hessian assembly of the rigid and
// SLAC modes.  TEST INFRASTRUCTURE ONLY: used by tests/, never by the product.
//
//   Point, GetCoordinate, UpdatePose, UpdateNormal, UpdatePoint, UpdateAllPointPN   FragmentOptimizer/PointCloud.h:6-176
//   rigid assembly (12-entry bucket, AddHessian + AddJb)                           FragmentOptimizer/OptApp.cpp:312-375,
//                                                                                  HashSparseMatrix.cpp:16-48
//   SLAC assembly (12 + 24 + 24 bucket into a dense upper-triangular matrix)       FragmentOptimizer/OptApp.cpp:473-560
//
// Plain scalar C++ (no Eigen), sequential sums in the reference's loop order.  Pinned by tests/test_fopt_oracle.py: the
// float32 point state bit for bit against oracle/_ref/libref_fopt.so (the reference's own PointCloud.{h,cpp} compiled in
// place), the assembled systems against what the reference PROGRAM (oracle/_ref/FragmentOptimizer_ref) hands to its solver
// (1e-12 rigid, 1e-10 SLAC, 1e-9 non-rigid: Eigen's fixed-size dot products add in another order).  The CHOLMOD solve and
// the regularizer (a few thousand lattice vertices) stay on the host and are not restated here.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <vector>

namespace {

struct Point {               // PointCloud.h:6-13
  int idx[8];
  float n[3];
  float val[8];
  float nval[8];
  float p[3];
};

struct Cloud {
  std::vector<Point> pts;
};

struct Pair {
  int i, j;
  std::vector<int> first, second;
};

struct Fopt {
  int num, resolution;
  float length, unit_length;
  int nper;
  std::vector<Cloud> clouds;
  std::vector<Pair> pairs;
};

inline int get_index(int res, int i, int j, int k) { return i + j * (res + 1) + k * (res + 1) * (res + 1); }   // PointCloud.h:54-56

// PointCloud::GetCoordinate, PointCloud.h:92-176 (float32 throughout; pt[3..5] are divided by unit_length_ in place)
bool get_coordinate(int res, float ul, const float* in6, Point& pt_out) {
  float pt[6];
  memcpy(pt, in6, sizeof pt);
  Point& point = pt_out;
  point.p[0] = pt[0];
  point.p[1] = pt[1];
  point.p[2] = pt[2];
  int corner[3] = {(int)floor(pt[0] / ul), (int)floor(pt[1] / ul), (int)floor(pt[2] / ul)};
  if (corner[0] < 0 || corner[0] >= res || corner[1] < 0 || corner[1] >= res || corner[2] < 0 || corner[2] >= res) return false;
  float r[3] = {pt[0] / ul - corner[0], pt[1] / ul - corner[1], pt[2] / ul - corner[2]};
  for (int t = 0; t < 8; t++)
    point.idx[t] = get_index(res, corner[0] + ((t >> 2) & 1), corner[1] + ((t >> 1) & 1), corner[2] + (t & 1)) * 3;
  point.val[0] = (1 - r[0]) * (1 - r[1]) * (1 - r[2]);
  point.val[1] = (1 - r[0]) * (1 - r[1]) * (r[2]);
  point.val[2] = (1 - r[0]) * (r[1]) * (1 - r[2]);
  point.val[3] = (1 - r[0]) * (r[1]) * (r[2]);
  point.val[4] = (r[0]) * (1 - r[1]) * (1 - r[2]);
  point.val[5] = (r[0]) * (1 - r[1]) * (r[2]);
  point.val[6] = (r[0]) * (r[1]) * (1 - r[2]);
  point.val[7] = (r[0]) * (r[1]) * (r[2]);
  pt[3] /= ul;
  pt[4] /= ul;
  pt[5] /= ul;
  point.nval[0] = -pt[3] * (1 - r[1]) * (1 - r[2]) - pt[4] * (1 - r[0]) * (1 - r[2]) - pt[5] * (1 - r[0]) * (1 - r[1]);
  point.nval[1] = -pt[3] * (1 - r[1]) * (r[2]) - pt[4] * (1 - r[0]) * (r[2]) + pt[5] * (1 - r[0]) * (1 - r[1]);
  point.nval[2] = -pt[3] * (r[1]) * (1 - r[2]) + pt[4] * (1 - r[0]) * (1 - r[2]) - pt[5] * (1 - r[0]) * (r[1]);
  point.nval[3] = -pt[3] * (r[1]) * (r[2]) + pt[4] * (1 - r[0]) * (r[2]) + pt[5] * (1 - r[0]) * (r[1]);
  point.nval[4] = pt[3] * (1 - r[1]) * (1 - r[2]) - pt[4] * (r[0]) * (1 - r[2]) - pt[5] * (r[0]) * (1 - r[1]);
  point.nval[5] = pt[3] * (1 - r[1]) * (r[2]) - pt[4] * (r[0]) * (r[2]) + pt[5] * (r[0]) * (1 - r[1]);
  point.nval[6] = pt[3] * (r[1]) * (1 - r[2]) + pt[4] * (r[0]) * (1 - r[2]) - pt[5] * (r[0]) * (r[1]);
  point.nval[7] = pt[3] * (r[1]) * (r[2]) + pt[4] * (r[0]) * (r[2]) + pt[5] * (r[0]) * (r[1]);
  point.n[0] = pt[3];
  point.n[1] = pt[4];
  point.n[2] = pt[5];
  return true;
}

// PointCloud::UpdatePose, PointCloud.h:71-83: Matrix4f * Vector4f per row ((m0*x + m1*y) + m2*z) + m3*w (Eigen coefficient order)
void update_pose(Cloud& c, const float* M) {
  for (Point& q : c.pts) {
    float p[3], n[3];
    for (int r = 0; r < 3; r++) {
      p[r] = ((M[4 * r] * q.p[0] + M[4 * r + 1] * q.p[1]) + M[4 * r + 2] * q.p[2]) + M[4 * r + 3] * 1.0f;
      n[r] = ((M[4 * r] * q.n[0] + M[4 * r + 1] * q.n[1]) + M[4 * r + 2] * q.n[2]) + M[4 * r + 3] * 0.0f;
    }
    memcpy(q.p, p, sizeof p);
    memcpy(q.n, n, sizeof n);
  }
}

// PointCloud::UpdateAllPointPN, PointCloud.h:44-52 (UpdateNormal :58-69, UpdatePoint :85-94); ctr = this fragment's slice
void update_point_pn(Cloud& c, const double* ctr) {
  for (Point& q : c.pts) {
    for (int i = 0; i < 3; i++) {
      q.n[i] = 0.0f;
      for (int j = 0; j < 8; j++) q.n[i] += q.nval[j] * (float)ctr[q.idx[j] + i];
    }
    const float len = (float)sqrt((double)(q.n[0] * q.n[0] + q.n[1] * q.n[1] + q.n[2] * q.n[2]));
    q.n[0] /= len;
    q.n[1] /= len;
    q.n[2] /= len;
    for (int i = 0; i < 3; i++) {
      double pos = 0.0;
      for (int j = 0; j < 8; j++) pos += q.val[j] * (float)ctr[q.idx[j] + i];   // float * float, widened for the sum
      q.p[i] = (float)pos;
    }
  }
}

// PointCloud::UpdateAllNormal, PointCloud.h:32-36 (UpdateNormal :58-69): normals only (non-rigid mode, OptApp.cpp:151-153)
void update_normals(Cloud& c, const double* ctr) {
  for (Point& q : c.pts) {
    for (int i = 0; i < 3; i++) {
      q.n[i] = 0.0f;
      for (int j = 0; j < 8; j++) q.n[i] += q.nval[j] * (float)ctr[q.idx[j] + i];
    }
    const float len = (float)sqrt((double)(q.n[0] * q.n[0] + q.n[1] * q.n[1] + q.n[2] * q.n[2]));
    q.n[0] /= len;
    q.n[1] /= len;
    q.n[2] /= len;
  }
}

// The two 24-entry buckets of the non-rigid mode, OptApp.cpp:176-190: entry c*8 + t  (c = x,y,z component, t = vertex)
void nonrigid_bucket(const Point& pi, const Point& pj, double weight, int idx1[24], double val1[24], int idx2[24], double val2[24]) {
  for (int t = 0; t < 8; t++)
    for (int c = 0; c < 3; c++) {
      idx1[c * 8 + t] = pi.idx[t] + c;
      val1[c * 8 + t] = pi.val[t] * weight * pi.n[c];
      idx2[c * 8 + t] = pj.idx[t] + c;
      val2[c * 8 + t] = -pj.val[t] * weight * pi.n[c];
    }
}

// The 12 pose entries of the rigid bucket and b, OptApp.cpp:337-365 (Vector4d dots written out; zero terms kept out).
void rigid_bucket(const Point& pi, const Point& pj, double val[12], double& b) {
  const double ppi[3] = {pi.p[0], pi.p[1], pi.p[2]}, ppj[3] = {pj.p[0], pj.p[1], pj.p[2]}, npi[3] = {pi.n[0], pi.n[1], pi.n[2]};
  const double d[3] = {ppi[0] - ppj[0], ppi[1] - ppj[1], ppi[2] - ppj[2]};
  b = (d[0] * npi[0] + d[1] * npi[1]) + d[2] * npi[2];
  val[0] = (-ppi[2] * npi[1] + ppi[1] * npi[2]) + (-npi[2] * d[1] + npi[1] * d[2]);
  val[1] = (ppi[2] * npi[0] - ppi[0] * npi[2]) + (npi[2] * d[0] - npi[0] * d[2]);
  val[2] = (-ppi[1] * npi[0] + ppi[0] * npi[1]) + (-npi[1] * d[0] + npi[0] * d[1]);
  val[3] = npi[0];
  val[4] = npi[1];
  val[5] = npi[2];
  val[6] = -(-ppj[2] * npi[1] + ppj[1] * npi[2]);
  val[7] = -(ppj[2] * npi[0] - ppj[0] * npi[2]);
  val[8] = -(-ppj[1] * npi[0] + ppj[0] * npi[1]);
  val[9] = -npi[0];
  val[10] = -npi[1];
  val[11] = -npi[2];
}

// The 60-entry SLAC bucket, OptApp.cpp:489-535.  rot_t = pose_rot_t_[.] row-major 3x3.
void slac_bucket(const Point& pi, const Point& pj, int i, int j, int num, const double* rot_t_i, const double* rot_t_j, int idx[60],
                 double val[60], double& b) {
  const double ppi[3] = {pi.p[0], pi.p[1], pi.p[2]}, ppj[3] = {pj.p[0], pj.p[1], pj.p[2]}, npi[3] = {pi.n[0], pi.n[1], pi.n[2]};
  const double d[3] = {ppi[0] - ppj[0], ppi[1] - ppj[1], ppi[2] - ppj[2]};
  b = (d[0] * npi[0] + d[1] * npi[1]) + d[2] * npi[2];
  for (int q = 0; q < 6; q++) {
    idx[q] = i * 6 + q;
    idx[6 + q] = j * 6 + q;
  }
  const double t[3] = {ppj[1] * npi[2] - ppj[2] * npi[1], ppj[2] * npi[0] - ppj[0] * npi[2], ppj[0] * npi[1] - ppj[1] * npi[0]};   // ppj x npi
  for (int q = 0; q < 3; q++) {
    val[q] = t[q];
    val[3 + q] = npi[q];
    val[6 + q] = -t[q];
    val[9 + q] = -npi[q];
  }
  double dTi[3], dTj[3];
  for (int r = 0; r < 3; r++) {
    dTi[r] = (rot_t_i[3 * r] * npi[0] + rot_t_i[3 * r + 1] * npi[1]) + rot_t_i[3 * r + 2] * npi[2];
    dTj[r] = -((rot_t_j[3 * r] * npi[0] + rot_t_j[3 * r + 1] * npi[1]) + rot_t_j[3 * r + 2] * npi[2]);
  }
  for (int ll = 0; ll < 8; ll++)
    for (int xyz = 0; xyz < 3; xyz++) {
      idx[12 + ll * 3 + xyz] = 6 * num + pi.idx[ll] + xyz;
      val[12 + ll * 3 + xyz] = pi.val[ll] * dTi[xyz];
      idx[36 + ll * 3 + xyz] = 6 * num + pj.idx[ll] + xyz;
      val[36 + ll * 3 + xyz] = pj.val[ll] * dTj[xyz];
    }
}

}  // namespace

extern "C" {

void* fopt_create(int num, int resolution, float length) {
  Fopt* f = new Fopt();
  f->num = num;
  f->resolution = resolution;
  f->length = length;
  f->unit_length = length / resolution;                       // PointCloud.cpp:10 (float / int)
  f->nper = (resolution + 1) * (resolution + 1) * (resolution + 1) * 3;
  f->clouds.resize((size_t)num);
  return f;
}
void fopt_destroy(void* h) { delete static_cast<Fopt*>(h); }

// LoadFromXYZNFile / LoadFromPCDFile body (PointCloud.cpp:22-63).  Returns -1 on success, else the index of the first
// out-of-bound point (the reference prints "Point out of bound" and stops loading there).
int fopt_set_cloud(void* h, int frag, const float* xyz, const float* nrm, int n) {
  Fopt& f = *static_cast<Fopt*>(h);
  Cloud& c = f.clouds[(size_t)frag];
  c.pts.clear();
  for (int k = 0; k < n; k++) {
    float x[6] = {xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2], nrm[3 * k], nrm[3 * k + 1], nrm[3 * k + 2]};
    c.pts.resize(c.pts.size() + 1);
    if (!get_coordinate(f.resolution, f.unit_length, x, c.pts.back())) return k;
  }
  return -1;
}

int fopt_cloud_size(void* h, int frag) { return (int)static_cast<Fopt*>(h)->clouds[(size_t)frag].pts.size(); }

// state read-back: idx0 (= idx_[0]), val_[8], nval_[8], p_[3], n_[3] per point
void fopt_get_points(void* h, int frag, int* idx0, float* val, float* nval, float* p, float* n) {
  const Cloud& c = static_cast<Fopt*>(h)->clouds[(size_t)frag];
  for (size_t k = 0; k < c.pts.size(); k++) {
    if (idx0) idx0[k] = c.pts[k].idx[0];
    if (val) memcpy(val + 8 * k, c.pts[k].val, 8 * sizeof(float));
    if (nval) memcpy(nval + 8 * k, c.pts[k].nval, 8 * sizeof(float));
    if (p) memcpy(p + 3 * k, c.pts[k].p, 3 * sizeof(float));
    if (n) memcpy(n + 3 * k, c.pts[k].n, 3 * sizeof(float));
  }
}

void fopt_update_pose(void* h, int frag, const float* M16) { update_pose(static_cast<Fopt*>(h)->clouds[(size_t)frag], M16); }
void fopt_update_point_pn(void* h, int frag, const double* ctr_slice) { update_point_pn(static_cast<Fopt*>(h)->clouds[(size_t)frag], ctr_slice); }

void fopt_clear_pairs(void* h) { static_cast<Fopt*>(h)->pairs.clear(); }
void fopt_add_pair(void* h, int i, int j, const int* pairs2, int n) {       // corres_<i>_<j>.txt rows (first, second), OptApp.h:23-35
  Fopt& f = *static_cast<Fopt*>(h);
  Pair p;
  p.i = i;
  p.j = j;
  p.first.resize((size_t)n);
  p.second.resize((size_t)n);
  for (int k = 0; k < n; k++) {
    p.first[(size_t)k] = pairs2[2 * k];
    p.second[(size_t)k] = pairs2[2 * k + 1];
  }
  f.pairs.push_back(p);
}

// OptimizeRigid's assembly, OptApp.cpp:312-375: JJ dense (6 num)^2 row-major, FULL symmetric (AddHessian adds both
// triangles) including the "+1" each pair puts on the first six diagonal entries (:322-324); Jb; score = sum b^2.
void fopt_assemble_rigid(void* h, double* JJ, double* Jb, double* score) {
  Fopt& f = *static_cast<Fopt*>(h);
  const int N = 6 * f.num;
  memset(JJ, 0, sizeof(double) * (size_t)N * N);
  memset(Jb, 0, sizeof(double) * (size_t)N);
  double total = 0.0;
  for (const Pair& pr : f.pairs) {
    for (int k = 0; k < 6; k++) JJ[(size_t)k * N + k] += 1.0;
    int idx[12];
    for (int q = 0; q < 6; q++) {
      idx[q] = pr.i * 6 + q;
      idx[6 + q] = pr.j * 6 + q;
    }
    double sc = 0.0;
    for (size_t k = 0; k < pr.first.size(); k++) {
      const Point& pi = f.clouds[(size_t)pr.i].pts[(size_t)pr.first[k]];
      const Point& pj = f.clouds[(size_t)pr.j].pts[(size_t)pr.second[k]];
      double val[12], b;
      rigid_bucket(pi, pj, val, b);
      sc += b * b;
      for (int a = 0; a < 12; a++) {
        for (int c = 0; c < 12; c++) JJ[(size_t)idx[a] * N + idx[c]] += val[a] * val[c];
        Jb[idx[a]] += val[a] * b;
      }
    }
    total += sc;
  }
  *score = total;
}

// OptimizeSLAC's data term, OptApp.cpp:473-560: dense (6 num + nper)^2 row-major, UPPER triangle (:537-548), Jb, score.
// pose_rot_t: num * 9 doubles (row-major pose_[l].block<3,3>(0,0).transpose()).
void fopt_assemble_slac(void* h, const double* pose_rot_t, double* JJ, double* Jb, double* score) {
  Fopt& f = *static_cast<Fopt*>(h);
  const int N = 6 * f.num + f.nper;
  memset(JJ, 0, sizeof(double) * (size_t)N * N);
  memset(Jb, 0, sizeof(double) * (size_t)N);
  double total = 0.0;
  for (const Pair& pr : f.pairs) {
    double sc = 0.0;
    for (size_t k = 0; k < pr.first.size(); k++) {
      const Point& pi = f.clouds[(size_t)pr.i].pts[(size_t)pr.first[k]];
      const Point& pj = f.clouds[(size_t)pr.j].pts[(size_t)pr.second[k]];
      int idx[60];
      double val[60], b;
      slac_bucket(pi, pj, pr.i, pr.j, f.num, pose_rot_t + 9 * pr.i, pose_rot_t + 9 * pr.j, idx, val, b);
      sc += b * b;
      for (int a = 0; a < 60; a++) {
        JJ[(size_t)idx[a] * N + idx[a]] += val[a] * val[a];
        for (int c = a + 1; c < 60; c++) {
          if (idx[a] == idx[c]) JJ[(size_t)idx[a] * N + idx[c]] += 2 * val[a] * val[c];
          else if (idx[a] < idx[c]) JJ[(size_t)idx[a] * N + idx[c]] += val[a] * val[c];
          else JJ[(size_t)idx[c] * N + idx[a]] += val[a] * val[c];
        }
        Jb[idx[a]] += b * val[a];
      }
    }
    total += sc;
  }
  *score = total;
}

void fopt_update_normals(void* h, int frag, const double* ctr_slice) { update_normals(static_cast<Fopt*>(h)->clouds[(size_t)frag], ctr_slice); }

// OptimizeNonrigid's data term, OptApp.cpp:159-206: thisAA - baseAA as merged triplets (row, col, value), rows/cols global
// (fragment * nper + lattice index).  mati / matj add full 24x24 blocks on the diagonal fragment blocks, matij the (i, j)
// block only (HashSparseMatrix.cpp:42-66).  Call with rows == NULL to get the count.
static std::map<long long, double>* g_tri = nullptr;
long fopt_assemble_nonrigid(void* h, double weight, long long* keys, double* vals) {
  Fopt& f = *static_cast<Fopt*>(h);
  if (!keys) {
    delete g_tri;
    g_tri = new std::map<long long, double>();
    const long long M = (long long)f.nper * f.num;
    for (const Pair& pr : f.pairs) {
      const long long oi = (long long)pr.i * f.nper, oj = (long long)pr.j * f.nper;
      for (size_t k = 0; k < pr.first.size(); k++) {
        const Point& pi = f.clouds[(size_t)pr.i].pts[(size_t)pr.first[k]];
        const Point& pj = f.clouds[(size_t)pr.j].pts[(size_t)pr.second[k]];
        int idx1[24], idx2[24];
        double val1[24], val2[24];
        nonrigid_bucket(pi, pj, weight, idx1, val1, idx2, val2);
        for (int a = 0; a < 24; a++)
          for (int c = 0; c < 24; c++) {
            (*g_tri)[(oi + idx1[a]) * M + (oi + idx1[c])] += val1[a] * val1[c];
            (*g_tri)[(oj + idx2[a]) * M + (oj + idx2[c])] += val2[a] * val2[c];
            (*g_tri)[(oi + idx1[a]) * M + (oj + idx2[c])] += val1[a] * val2[c];
          }
      }
    }
    return (long)g_tri->size();
  }
  long n = 0;
  for (const auto& kv : *g_tri) {
    keys[n] = kv.first;
    vals[n] = kv.second;
    n++;
  }
  return n;
}

void fopt_nonrigid_bucket(void* h, int i, int ii, int j, int jj, double weight, int* idx1, double* val1, int* idx2, double* val2) {
  Fopt& f = *static_cast<Fopt*>(h);
  nonrigid_bucket(f.clouds[(size_t)i].pts[(size_t)ii], f.clouds[(size_t)j].pts[(size_t)jj], weight, idx1, val1, idx2, val2);
}

// single-correspondence buckets for the pin test
void fopt_rigid_bucket(void* h, int i, int ii, int j, int jj, double* val12, double* b) {
  Fopt& f = *static_cast<Fopt*>(h);
  rigid_bucket(f.clouds[(size_t)i].pts[(size_t)ii], f.clouds[(size_t)j].pts[(size_t)jj], val12, *b);
}
void fopt_slac_bucket(void* h, int i, int ii, int j, int jj, const double* pose_rot_t, int* idx60, double* val60, double* b) {
  Fopt& f = *static_cast<Fopt*>(h);
  slac_bucket(f.clouds[(size_t)i].pts[(size_t)ii], f.clouds[(size_t)j].pts[(size_t)jj], i, j, f.num, pose_rot_t + 9 * i, pose_rot_t + 9 * j, idx60,
              val60, *b);
}

}  // extern "C"
