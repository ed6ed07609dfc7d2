// icp_oracle.cpp -- TEST INFRASTRUCTURE ONLY.  CPU restatement of path B of the reference
// (qianyizh/ElasticReconstruction, BuildCorrespondence/CorresApp.cpp:112-319): the inlier pre-check,
// PCL's point-to-plane ICP as configured there, FindCorrespondence and the information matrix.
// Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the CHECKER; the
// product never includes, links or calls anything in oracle/.
//
// *** PARITY UNPINNED. ***  The arithmetic of the ICP step lives in PCL 1.7 (the author's fork
// qianyizh/StanfordPCL, unversioned; BuildCorrespondence.vcxproj:109,128,137; README.txt:7-8) and
// FLANN.  Neither is vendored in /root/reference nor installed here, there is no network, and the
// reference holds no test or golden vector for this path.  What follows restates PCL 1.7's
// PUBLISHED algorithm (pcl::IterativeClosestPoint::computeTransformation,
// registration::CorrespondenceEstimation::determineCorrespondences,
// registration::TransformationEstimationPointToPlaneLLS::estimateRigidTransformation,
// registration::DefaultConvergenceCriteria::hasConverged, pcl::transformPointCloudWithNormals,
// KdTreeFLANN::nearestKSearch with FLANN L2_Simple) as listed in SURVEY.md Appendix B; every
// assumption is marked [PCL].  Everything OUTSIDE PCL follows CorresApp.cpp line by line and is cited.
// Pins that exist: synthetic pairs with known ground truth, scipy.spatial.cKDTree as an independent
// exact-NN check, and the known-answer STRUCTURE of the information matrix visible in
// Matlab_Toolbox/Example/Data/RegistrationEvaluation/*/gt.info (N*I3 block, antisymmetric +-2*sum(s)).
//
// Nearest neighbours: exact 1-NN on xyz inside a cutoff radius, found through a uniform grid whose
// cell edge is >= the radius (27-cell search).  Every consumer in the reference discards matches beyond
// its cutoff (CorresApp.cpp:154,260; PCL max correspondence distance), so "no neighbour within the
// cutoff" is all they need.  Squared distance is float32 ((dx*dx) + dy*dy) + dz*dz [PCL: FLANN
// L2_Simple]; ties are broken towards the lower target index [PCL: unspecified in FLANN].
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct Cloud {
  int n = 0;
  std::vector<float> xyz, nrm;        // AoS, 3 floats per point (pointclouds_[i] after the NaN filter, CorresApp.cpp:94-98)
  // uniform grid over the points (built lazily for a given cell size)
  float cell = 0.f;
  float org[3] = {0, 0, 0};
  int dim[3] = {0, 0, 0};
  std::vector<int> cell_start;        // size ncells + 1
  std::vector<int> order;             // point indices sorted by cell
};

void build_grid(Cloud& c, float cell) {
  if (c.cell == cell && !c.cell_start.empty()) return;
  c.cell = cell;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int i = 0; i < c.n; i++)
    for (int a = 0; a < 3; a++) {
      lo[a] = std::min(lo[a], c.xyz[3 * i + a]);
      hi[a] = std::max(hi[a], c.xyz[3 * i + a]);
    }
  if (c.n == 0) { lo[0] = lo[1] = lo[2] = 0; hi[0] = hi[1] = hi[2] = 0; }
  for (;;) {
    long total = 1;
    for (int a = 0; a < 3; a++) {
      c.org[a] = lo[a];
      c.dim[a] = (int)std::floor((hi[a] - lo[a]) / c.cell) + 1;
      total *= c.dim[a];
    }
    if (total <= (1L << 24)) break;
    c.cell *= 2.f;
  }
  const int ncell = c.dim[0] * c.dim[1] * c.dim[2];
  std::vector<int> id(c.n);
  c.cell_start.assign(ncell + 1, 0);
  for (int i = 0; i < c.n; i++) {
    int q[3];
    for (int a = 0; a < 3; a++) {
      q[a] = (int)std::floor((c.xyz[3 * i + a] - c.org[a]) / c.cell);
      q[a] = std::min(std::max(q[a], 0), c.dim[a] - 1);
    }
    id[i] = (q[2] * c.dim[1] + q[1]) * c.dim[0] + q[0];
    c.cell_start[id[i] + 1]++;
  }
  for (int k = 0; k < ncell; k++) c.cell_start[k + 1] += c.cell_start[k];
  c.order.resize(c.n);
  std::vector<int> fill(c.cell_start.begin(), c.cell_start.end() - 1);
  for (int i = 0; i < c.n; i++) c.order[fill[id[i]]++] = i;   // ascending index inside each cell
}

// Exact nearest neighbour of q among the points within `radius` (radius <= grid cell).  Returns index or -1.
int nearest(const Cloud& c, const float q[3], float radius, float* sqdist) {
  int qc[3];
  for (int a = 0; a < 3; a++) {
    float t = std::floor((q[a] - c.org[a]) / c.cell);
    if (!(t >= -1.f && t <= (float)c.dim[a])) return -1;             // farther than one cell from the grid
    qc[a] = (int)t;
  }
  int best = -1;
  float bestd = FLT_MAX;
  for (int dz = -1; dz <= 1; dz++) {
    int z = qc[2] + dz;
    if (z < 0 || z >= c.dim[2]) continue;
    for (int dy = -1; dy <= 1; dy++) {
      int y = qc[1] + dy;
      if (y < 0 || y >= c.dim[1]) continue;
      int x0 = std::max(qc[0] - 1, 0), x1 = std::min(qc[0] + 1, c.dim[0] - 1);
      if (x0 > x1) continue;
      int row = (z * c.dim[1] + y) * c.dim[0];
      for (int s = c.cell_start[row + x0]; s < c.cell_start[row + x1 + 1]; s++) {
        int i = c.order[s];
        float dx = q[0] - c.xyz[3 * i], dy2 = q[1] - c.xyz[3 * i + 1], dz2 = q[2] - c.xyz[3 * i + 2];
        float d = ((dx * dx) + dy2 * dy2) + dz2 * dz2;                // [PCL] FLANN L2_Simple, float32
        if (d < bestd || (d == bestd && i < best)) { bestd = d; best = i; }
      }
    }
  }
  if (best < 0) return -1;
  // The grid guarantees exactness only inside the radius; anything beyond is reported as "none".
  if (!((double)bestd <= (double)radius * (double)radius)) return -1;
  *sqdist = bestd;
  return best;
}

// [PCL] pcl::transformPointCloudWithNormals( in, out, Matrix4d ): evaluated in the matrix scalar type
// (double), stored as float;  p' = ((m00*x + m01*y) + m02*z) + m03,  n' = (m00*nx + m01*ny) + m02*nz.
void transform_double(const Cloud& src, const double* T, std::vector<float>& xyz, std::vector<float>* nrm) {
  xyz.resize(3 * (size_t)src.n);
  if (nrm) nrm->resize(3 * (size_t)src.n);
  for (int k = 0; k < src.n; k++) {
    double x = src.xyz[3 * k], y = src.xyz[3 * k + 1], z = src.xyz[3 * k + 2];
    for (int r = 0; r < 3; r++) xyz[3 * k + r] = (float)(((T[4 * r] * x + T[4 * r + 1] * y) + T[4 * r + 2] * z) + T[4 * r + 3]);
    if (nrm) {
      double nx = src.nrm[3 * k], ny = src.nrm[3 * k + 1], nz = src.nrm[3 * k + 2];
      for (int r = 0; r < 3; r++) (*nrm)[3 * k + r] = (float)((T[4 * r] * nx + T[4 * r + 1] * ny) + T[4 * r + 2] * nz);
    }
  }
}

// 6x6 solve, LU with partial pivoting, float64.  [PCL] uses Eigen's ATA.inverse() * ATb (PartialPivLU).
bool solve6(double A[36], double b[6], double x[6]) {
  int p[6] = {0, 1, 2, 3, 4, 5};
  for (int c = 0; c < 6; c++) {
    int piv = c;
    for (int r = c + 1; r < 6; r++)
      if (std::fabs(A[p[r] * 6 + c]) > std::fabs(A[p[piv] * 6 + c])) piv = r;
    std::swap(p[c], p[piv]);
    double d = A[p[c] * 6 + c];
    if (d == 0.0 || !std::isfinite(d)) return false;
    for (int r = c + 1; r < 6; r++) {
      double f = A[p[r] * 6 + c] / d;
      A[p[r] * 6 + c] = f;
      for (int k = c + 1; k < 6; k++) A[p[r] * 6 + k] -= f * A[p[c] * 6 + k];
    }
  }
  double y[6];
  for (int r = 0; r < 6; r++) {
    double s = b[p[r]];
    for (int k = 0; k < r; k++) s -= A[p[r] * 6 + k] * y[k];
    y[r] = s;
  }
  for (int r = 5; r >= 0; r--) {
    double s = y[r];
    for (int k = r + 1; k < 6; k++) s -= A[p[r] * 6 + k] * x[k];
    x[r] = s / A[p[r] * 6 + r];
  }
  return true;
}

// [PCL] TransformationEstimationPointToPlaneLLS::constructTransformationMatrix: Rz(gamma)*Ry(beta)*Rx(alpha)
// with full trigonometry in double, stored as float.
void construct_transform(const double x[6], float M[16]) {
  const double alpha = x[0], beta = x[1], gamma = x[2];
  for (int i = 0; i < 16; i++) M[i] = 0.f;
  M[0] = (float)(cos(gamma) * cos(beta));
  M[1] = (float)(-sin(gamma) * cos(alpha) + cos(gamma) * sin(beta) * sin(alpha));
  M[2] = (float)(sin(gamma) * sin(alpha) + cos(gamma) * sin(beta) * cos(alpha));
  M[4] = (float)(sin(gamma) * cos(beta));
  M[5] = (float)(cos(gamma) * cos(alpha) + sin(gamma) * sin(beta) * sin(alpha));
  M[6] = (float)(-cos(gamma) * sin(alpha) + sin(gamma) * sin(beta) * cos(alpha));
  M[8] = (float)(-sin(beta));
  M[9] = (float)(cos(beta) * sin(alpha));
  M[10] = (float)(cos(beta) * cos(alpha));
  M[3] = (float)x[3];
  M[7] = (float)x[4];
  M[11] = (float)x[5];
  M[15] = 1.f;
}

// float 4x4 product, coefficient order ((a0*b0 + a1*b1) + a2*b2) + a3*b3 [PCL/Eigen: final = delta * final]
void mat4f_mul(const float* A, const float* B, float* C) {
  float t[16];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++)
      t[r * 4 + c] = ((A[r * 4] * B[c] + A[r * 4 + 1] * B[4 + c]) + A[r * 4 + 2] * B[8 + c]) + A[r * 4 + 3] * B[12 + c];
  memcpy(C, t, sizeof t);
}

// [PCL] IterativeClosestPoint::transformCloud: pt_t = tr * pt in float32, per row ((m0*x + m1*y) + m2*z) + m3.
void transform_float_inplace(std::vector<float>& X, int n, const float* M) {
  for (int k = 0; k < n; k++) {
    float x = X[3 * k], y = X[3 * k + 1], z = X[3 * k + 2];
    for (int r = 0; r < 3; r++) X[3 * k + r] = ((M[4 * r] * x + M[4 * r + 1] * y) + M[4 * r + 2] * z) + M[4 * r + 3];
  }
}

}  // namespace

extern "C" {

void* icp_cloud_create(const float* xyz, const float* nrm, int n, float grid_cell) {
  Cloud* c = new Cloud();
  c->n = n;
  c->xyz.assign(xyz, xyz + 3 * (size_t)n);
  c->nrm.assign(nrm, nrm + 3 * (size_t)n);
  build_grid(*c, grid_cell * 1.001f);   // strictly larger than any admissible radius (float-rounding safety)
  return c;
}
void icp_cloud_destroy(void* c) { delete static_cast<Cloud*>(c); }
int icp_cloud_size(void* c) { return static_cast<Cloud*>(c)->n; }

// Raw NN pass for cross-checks: idx_out[k] = NN index of T*src[k] in tgt within max_dist, or -1.
void icp_nn_pass(void* src_, void* tgt_, const double* T, double max_dist, int* idx_out, float* sqd_out) {
  Cloud& src = *static_cast<Cloud*>(src_);
  Cloud& tgt = *static_cast<Cloud*>(tgt_);
  std::vector<float> X;
  transform_double(src, T, X, nullptr);
#pragma omp parallel for schedule(dynamic, 1024)
  for (int k = 0; k < src.n; k++) {
    float d = 0.f;
    idx_out[k] = nearest(tgt, &X[3 * (size_t)k], (float)max_dist, &d);
    if (sqd_out) sqd_out[k] = idx_out[k] >= 0 ? d : -1.f;
  }
}

// Registration pre-check, CorresApp.cpp:249-264: transformed = T * pcd1 (points+normals, Matrix4d);
// cnt = #{k : pointNKNSquaredDistance[0] < reg_dist_ * reg_dist_} (float promoted to double, strict <).
int icp_count_inliers(void* src_, void* tgt_, const double* T, double max_dist) {
  Cloud& src = *static_cast<Cloud*>(src_);
  Cloud& tgt = *static_cast<Cloud*>(tgt_);
  std::vector<float> X;
  transform_double(src, T, X, nullptr);
  int cnt = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : cnt)
  for (int k = 0; k < src.n; k++) {
    float d;
    int i = nearest(tgt, &X[3 * (size_t)k], (float)max_dist, &d);
    if (i >= 0 && (double)d < max_dist * max_dist) cnt++;
  }
  return cnt;
}

// icp.align( *transformed, guess ) as configured at CorresApp.cpp:295-306.  [PCL] throughout.
// stop_rule 0: PCL 1.7 DefaultConvergenceCriteria (max iterations; |rotation| and translation of the last
//              increment below the thresholds; absolute MSE change < 1e-12; relative MSE never, because
//              euclidean_fitness_epsilon_ defaults to -DBL_MAX); 1: PCL <= 1.6 (|sum(delta - delta_prev)| < eps).
int icp_align(void* src_, void* tgt_, const float* guess, double max_dist, int max_iter, double eps, int stop_rule,
              float* out, int* iterations, int* converged, double* fitness) {
  Cloud& src = *static_cast<Cloud*>(src_);
  Cloud& tgt = *static_cast<Cloud*>(tgt_);
  const int n = src.n;
  std::vector<float> X(src.xyz);                         // input_transformed
  float fin[16];                                         // final_transformation_ = guess
  memcpy(fin, guess, sizeof fin);
  bool ident = true;
  for (int i = 0; i < 16; i++) ident = ident && guess[i] == ((i % 5 == 0) ? 1.f : 0.f);
  if (!ident) transform_float_inplace(X, n, guess);      // transformCloud( *input_, *input_transformed, guess )
  float delta[16], prev_delta[16];
  for (int i = 0; i < 16; i++) delta[i] = prev_delta[i] = (i % 5 == 0) ? 1.f : 0.f;   // transformation_ = Identity
  int iter = 0;
  bool conv = false;
  double prev_mse = DBL_MAX;                             // correspondences_prev_mse_
  const double maxd2 = max_dist * max_dist;
  std::vector<int> nn(n);
  std::vector<float> nd(n);
  for (;;) {
    memcpy(prev_delta, delta, sizeof delta);             // previous_transformation_ = transformation_
    // determineCorrespondences: keep if distance <= max_dist^2 (no reciprocal test, no rejectors)
#pragma omp parallel for schedule(dynamic, 1024)
    for (int k = 0; k < n; k++) {
      float d = 0.f;
      int i = nearest(tgt, &X[3 * (size_t)k], (float)max_dist, &d);
      if (i >= 0 && !((double)d > maxd2)) { nn[k] = i; nd[k] = d; } else { nn[k] = -1; }
    }
    // estimateRigidTransformation: sums in double, in source-index order
    double ATA[36] = {0}, ATb[6] = {0};
    double mse = 0.0;
    long cnt = 0;
    for (int k = 0; k < n; k++) {
      if (nn[k] < 0) continue;
      const float sx = X[3 * k], sy = X[3 * k + 1], sz = X[3 * k + 2];
      const float* t = &tgt.xyz[3 * (size_t)nn[k]];
      const float* nr = &tgt.nrm[3 * (size_t)nn[k]];
      const float dx = t[0], dy = t[1], dz = t[2], nx = nr[0], ny = nr[1], nz = nr[2];
      cnt++;
      mse += (double)nd[k];
      if (!std::isfinite(sx) || !std::isfinite(sy) || !std::isfinite(sz) || !std::isfinite(nx) || !std::isfinite(ny) || !std::isfinite(nz)) continue;
      double a = nz * sy - ny * sz;                      // float32 expression widened to double [PCL]
      double b = nx * sz - nz * sx;
      double c = ny * sx - nx * sy;
      ATA[0] += a * a;  ATA[1] += a * b;  ATA[2] += a * c;  ATA[3] += a * nx;  ATA[4] += a * ny;  ATA[5] += a * nz;
      ATA[7] += b * b;  ATA[8] += b * c;  ATA[9] += b * nx; ATA[10] += b * ny; ATA[11] += b * nz;
      ATA[14] += c * c; ATA[15] += c * nx; ATA[16] += c * ny; ATA[17] += c * nz;
      ATA[21] += nx * nx; ATA[22] += nx * ny; ATA[23] += nx * nz;
      ATA[28] += ny * ny; ATA[29] += ny * nz;
      ATA[35] += nz * nz;
      double d = nx * dx + ny * dy + nz * dz - nx * sx - ny * sy - nz * sz;   // float32 expression [PCL]
      ATb[0] += a * d; ATb[1] += b * d; ATb[2] += c * d; ATb[3] += nx * d; ATb[4] += ny * d; ATb[5] += nz * d;
    }
    if (cnt < 3) { conv = false; break; }               // min_number_correspondences_ = 3
    for (int r = 0; r < 6; r++)
      for (int c2 = 0; c2 < r; c2++) ATA[r * 6 + c2] = ATA[c2 * 6 + r];
    double x[6];
    if (!solve6(ATA, ATb, x)) { conv = false; break; }
    construct_transform(x, delta);
    transform_float_inplace(X, n, delta);                // transformCloud( *input_transformed, *input_transformed, transformation_ )
    mat4f_mul(delta, fin, fin);                          // final_transformation_ = transformation_ * final_transformation_
    ++iter;
    // ---- convergence ----
    if (iter >= max_iter) { conv = true; break; }
    if (stop_rule == 0) {
      double cos_angle = 0.5 * (double)(delta[0] + delta[5] + delta[10] - 1.f);
      double tr2 = (double)(delta[3] * delta[3] + delta[7] * delta[7] + delta[11] * delta[11]);
      if (cos_angle >= 1.0 - eps && tr2 <= eps) { conv = true; break; }
      double cur = mse / (double)cnt;                    // calculateMSE: mean of the squared NN distances
      if (std::fabs(cur - prev_mse) < 1e-12) { conv = true; break; }
      prev_mse = cur;
    } else {
      float s = 0.f;
      for (int i = 0; i < 16; i++) s += delta[i] - prev_delta[i];
      if (std::fabs((double)s) < eps) { conv = true; break; }
    }
  }
  memcpy(out, fin, sizeof fin);
  if (iterations) *iterations = iter;
  if (converged) *converged = conv ? 1 : 0;
  if (fitness) {
    // [PCL] getFitnessScore(): mean squared NN distance of final*source to target -- logging only
    // (CorresApp.cpp:307).  Restricted here to neighbours within max_dist (grid search radius).
    std::vector<float> Y(src.xyz);
    transform_float_inplace(Y, n, fin);
    double s = 0.0;
    long m = 0;
    for (int k = 0; k < n; k++) {
      float d;
      if (nearest(tgt, &Y[3 * (size_t)k], (float)max_dist, &d) >= 0) { s += d; m++; }
    }
    *fitness = m ? s / (double)m : DBL_MAX;
  }
  return 0;
}

// FindCorrespondence, CorresApp.cpp:144-161 and :186-208.
int icp_find_correspondence(void* src_, void* tgt_, const double* T, double dist, double normal_cos, int* pairs,
                            int capacity, int* n_pairs, double* info36) {
  Cloud& src = *static_cast<Cloud*>(src_);
  Cloud& tgt = *static_cast<Cloud*>(tgt_);
  std::vector<float> X, N;
  transform_double(src, T, X, &N);                       // :145
  std::vector<int> nn(src.n);
#pragma omp parallel for schedule(dynamic, 1024)
  for (int k = 0; k < src.n; k++) {
    float d;
    int i = nearest(tgt, &X[3 * (size_t)k], (float)dist, &d);
    nn[k] = -1;
    if (i >= 0 && (double)d < dist * dist) {             // :154 strict <, float promoted to double
      // NormalDot, CorresApp.h:58-60: float32 products and sums, compared as double
      float dot = tgt.nrm[3 * (size_t)i] * N[3 * (size_t)k] + tgt.nrm[3 * (size_t)i + 1] * N[3 * (size_t)k + 1] +
                  tgt.nrm[3 * (size_t)i + 2] * N[3 * (size_t)k + 2];
      if ((double)dot > normal_cos) nn[k] = i;            // :155
    }
  }
  int m = 0;
  double ATA[36] = {0};
  for (int k = 0; k < src.n; k++) {
    if (nn[k] < 0) continue;
    if (m < capacity) { pairs[2 * m] = nn[k]; pairs[2 * m + 1] = k; }   // CorrespondencePair( nn, k ), :157
    m++;
    if (info36) {                                        // :192-204, s = UNtransformed pcd1 point
      const float sx = src.xyz[3 * k], sy = src.xyz[3 * k + 1], sz = src.xyz[3 * k + 2];
      const double A[3][6] = {{1, 0, 0, 0, (double)(2 * sz), (double)(-2 * sy)},
                              {0, 1, 0, (double)(-2 * sz), 0, (double)(2 * sx)},
                              {0, 0, 1, (double)(2 * sy), (double)(-2 * sx), 0}};
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) ATA[i * 6 + j] += (A[0][i] * A[0][j] + A[1][i] * A[1][j]) + A[2][i] * A[2][j];
    }
  }
  if (info36) memcpy(info36, ATA, sizeof ATA);
  *n_pairs = m;
  return m > capacity ? 1 : 0;
}

// RansacCurvature::getFitness, GlobalRegistration/RansacCurvature.h:661-704: input_transformed =
// transformPointCloud( *input_, final_transformation_ ) with a Matrix4f ([PCL] float32, per row ((m0*x + m1*y) + m2*z) + m3);
// inlier iff nn_dists[0] < corr_dist_threshold^2 (float compare, :670,:687); fitness = float32 running sum of the
// inlier distances in point order / inliers (FLT_MAX when none, :697-703).  Also returns the float64 sum for the
// order-independent comparison with the parallel implementation.
int icp_ransac_fitness(void* src_, void* tgt_, const float* M, float corr_dist_threshold, float* fitness_f32, double* sum_f64) {
  Cloud& src = *static_cast<Cloud*>(src_);
  Cloud& tgt = *static_cast<Cloud*>(tgt_);
  std::vector<float> X(src.xyz);
  transform_float_inplace(X, src.n, M);
  const float max_range = corr_dist_threshold * corr_dist_threshold;
  std::vector<float> dd((size_t)src.n);
  std::vector<char> in((size_t)src.n);
#pragma omp parallel for schedule(dynamic, 1024)
  for (int k = 0; k < src.n; k++) {
    float d;
    const int i = nearest(tgt, &X[3 * (size_t)k], corr_dist_threshold, &d);
    in[(size_t)k] = i >= 0 && d < max_range;
    dd[(size_t)k] = d;
  }
  int cnt = 0;
  float f = 0.0f;
  double s = 0.0;
  for (int k = 0; k < src.n; k++)
    if (in[(size_t)k]) {
      cnt++;
      f += dd[(size_t)k];
      s += (double)dd[(size_t)k];
    }
  if (fitness_f32) *fitness_f32 = cnt > 0 ? f / (float)cnt : FLT_MAX;
  if (sum_f64) *sum_f64 = s;
  return cnt;
}

// RansacCurvature::getFitness's lists (GlobalRegistration/RansacCurvature.h:680-695: inliers.push_back( i ),
// inliers_target.push_back( nn_indices[ 0 ] ) in point order) followed by getInformation (:707-733): for every inlier the
// 3x6 matrix A = [ I | 0 2z -2y ; -2z 0 2x ; 2y -2x 0 ] built from the FLOAT coordinates (2 * sz is a float product) widened
// to double, information += A^T A, once over the source inliers and once over their target matches.
// pairs (2 ints per inlier) must hold src.n entries.  Returns the inlier count.
int icp_ransac_inliers(void* src_, void* tgt_, const float* M, float corr_dist_threshold, int* pairs, double* info_source36,
                       double* info_target36) {
  Cloud& src = *static_cast<Cloud*>(src_);
  Cloud& tgt = *static_cast<Cloud*>(tgt_);
  std::vector<float> X(src.xyz);
  transform_float_inplace(X, src.n, M);
  const float max_range = corr_dist_threshold * corr_dist_threshold;
  std::vector<int> nn((size_t)src.n);
#pragma omp parallel for schedule(dynamic, 1024)
  for (int k = 0; k < src.n; k++) {
    float d;
    const int i = nearest(tgt, &X[3 * (size_t)k], corr_dist_threshold, &d);
    nn[(size_t)k] = (i >= 0 && d < max_range) ? i : -1;
  }
  int cnt = 0;
  for (int q = 0; q < 36; q++) info_source36[q] = info_target36[q] = 0.0;
  for (int k = 0; k < src.n; k++) {
    if (nn[(size_t)k] < 0) continue;
    pairs[2 * cnt] = k;
    pairs[2 * cnt + 1] = nn[(size_t)k];
    cnt++;
    for (int side = 0; side < 2; side++) {
      const float* p = side == 0 ? &src.xyz[3 * (size_t)k] : &tgt.xyz[3 * (size_t)nn[(size_t)k]];
      const float x2 = 2 * p[0], y2 = 2 * p[1], z2 = 2 * p[2];
      const double A[3][6] = {{1, 0, 0, 0, (double)z2, (double)-y2}, {0, 1, 0, (double)-z2, 0, (double)x2}, {0, 0, 1, (double)y2, (double)-x2, 0}};
      double* I = side == 0 ? info_source36 : info_target36;
      for (int r = 0; r < 6; r++)
        for (int c = 0; c < 6; c++) I[r * 6 + c] += (A[0][r] * A[0][c] + A[1][r] * A[1][c]) + A[2][r] * A[2][c];
    }
  }
  return cnt;
}

}  // extern "C"
