// ref_fopt_driver.cpp -- C-ABI over the reference's own FragmentOptimizer/PointCloud.h (compiled IN PLACE, unmodified,
// against the reference's vendored Eigen) plus the bucket expressions of OptApp.cpp written on the same Eigen types.
// Built into oracle/_ref/libref_fopt.so only where /root/reference exists; it pins oracle/fopt_oracle.cpp
// (tests/test_fopt_oracle.py).  OptApp.cpp itself cannot be compiled here: it needs CHOLMOD (SuiteSparse) through
// Eigen/CholmodSupport, and its matrices are locals of the Optimize* functions.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "PointCloud.h"            // the reference's header: Point, PointCloud::GetCoordinate / UpdatePose / UpdateAllPointPN
#include "external/Eigen/Geometry"

extern "C" {

void* rfopt_cloud_create(int index, int resolution, float length) { return new PointCloud(index, resolution, length); }
void rfopt_cloud_destroy(void* c) { delete static_cast<PointCloud*>(c); }

int rfopt_cloud_load(void* c_, const float* xyz, const float* nrm, int n) {   // LoadFromXYZNFile's loop body, PointCloud.cpp:50-58
  PointCloud& c = *static_cast<PointCloud*>(c_);
  c.points_.clear();
  for (int k = 0; k < n; k++) {
    float x[6] = {xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2], nrm[3 * k], nrm[3 * k + 1], nrm[3 * k + 2]};
    c.points_.resize(c.points_.size() + 1);
    if (c.GetCoordinate(x, c.points_.back()) == false) return k;
  }
  return -1;
}

void rfopt_get_points(void* c_, int* idx0, float* val, float* nval, float* p, float* n) {
  PointCloud& c = *static_cast<PointCloud*>(c_);
  for (size_t k = 0; k < c.points_.size(); k++) {
    idx0[k] = c.points_[k].idx_[0];
    memcpy(val + 8 * k, c.points_[k].val_, 8 * sizeof(float));
    memcpy(nval + 8 * k, c.points_[k].nval_, 8 * sizeof(float));
    memcpy(p + 3 * k, c.points_[k].p_, 3 * sizeof(float));
    memcpy(n + 3 * k, c.points_[k].n_, 3 * sizeof(float));
  }
}

void rfopt_update_pose(void* c_, const float* M16_rowmajor) {
  Eigen::Matrix4f M;
  for (int r = 0; r < 4; r++)
    for (int q = 0; q < 4; q++) M(r, q) = M16_rowmajor[4 * r + q];
  static_cast<PointCloud*>(c_)->UpdatePose(M);
}

// ctr_full: the whole expand_ctr vector (the cloud adds its own offset_)
void rfopt_update_point_pn(void* c_, const double* ctr_full, int size) {
  Eigen::VectorXd ctr(size);
  for (int i = 0; i < size; i++) ctr(i) = ctr_full[i];
  static_cast<PointCloud*>(c_)->UpdateAllPointPN(ctr);
}

void rfopt_update_normals(void* c_, const double* ctr_full, int size) {
  Eigen::VectorXd ctr(size);
  for (int i = 0; i < size; i++) ctr(i) = ctr_full[i];
  static_cast<PointCloud*>(c_)->UpdateAllNormal(ctr);
}

// OptApp.cpp:176-190 verbatim expressions (non-rigid buckets)
void rfopt_nonrigid_bucket(void* ci_, int ii, void* cj_, int jj, double weight_, int* idx1, double* val1, int* idx2, double* val2) {
  Point& pi = static_cast<PointCloud*>(ci_)->points_[ii];
  Point& pj = static_cast<PointCloud*>(cj_)->points_[jj];
  for (int t = 0; t < 8; t++) {
    idx1[t] = pi.idx_[t];
    val1[t] = pi.val_[t] * weight_ * pi.n_[0];
    idx1[8 + t] = pi.idx_[t] + 1;
    val1[8 + t] = pi.val_[t] * weight_ * pi.n_[1];
    idx1[16 + t] = pi.idx_[t] + 2;
    val1[16 + t] = pi.val_[t] * weight_ * pi.n_[2];
    idx2[t] = pj.idx_[t];
    val2[t] = -pj.val_[t] * weight_ * pi.n_[0];
    idx2[8 + t] = pj.idx_[t] + 1;
    val2[8 + t] = -pj.val_[t] * weight_ * pi.n_[1];
    idx2[16 + t] = pj.idx_[t] + 2;
    val2[16 + t] = -pj.val_[t] * weight_ * pi.n_[2];
  }
}

// OptApp.cpp:337-365 verbatim expressions (rigid bucket)
void rfopt_rigid_bucket(void* ci_, int ii, void* cj_, int jj, double* val, double* b_out) {
  Point& pi = static_cast<PointCloud*>(ci_)->points_[ii];
  Point& pj = static_cast<PointCloud*>(cj_)->points_[jj];
  Eigen::Vector4d ppi(pi.p_[0], pi.p_[1], pi.p_[2], 1.0);
  Eigen::Vector4d ppj(pj.p_[0], pj.p_[1], pj.p_[2], 1.0);
  Eigen::Vector4d npi(pi.n_[0], pi.n_[1], pi.n_[2], 0.0);
  double b = (ppi - ppj).dot(npi);
  val[0] = Eigen::Vector4d(0, -ppi(2), ppi(1), 1).dot(npi) + Eigen::Vector4d(0, -npi(2), npi(1), 0).dot(ppi - ppj);
  val[1] = Eigen::Vector4d(ppi(2), 0, -ppi(0), 1).dot(npi) + Eigen::Vector4d(npi(2), 0, -npi(0), 0).dot(ppi - ppj);
  val[2] = Eigen::Vector4d(-ppi(1), ppi(0), 0, 1).dot(npi) + Eigen::Vector4d(-npi(1), npi(0), 0, 0).dot(ppi - ppj);
  val[3] = npi(0);
  val[4] = npi(1);
  val[5] = npi(2);
  val[6] = -Eigen::Vector4d(0, -ppj(2), ppj(1), 1).dot(npi);
  val[7] = -Eigen::Vector4d(ppj(2), 0, -ppj(0), 1).dot(npi);
  val[8] = -Eigen::Vector4d(-ppj(1), ppj(0), 0, 1).dot(npi);
  val[9] = -npi(0);
  val[10] = -npi(1);
  val[11] = -npi(2);
  *b_out = b;
}

// OptApp.cpp:489-535 verbatim expressions (SLAC bucket); rot_t_* = pose_rot_t_[.] row-major
void rfopt_slac_bucket(void* ci_, int ii, int i, void* cj_, int jj, int j, int num_, const double* rot_t_i, const double* rot_t_j, int* idx,
                       double* val, double* b_out) {
  Point& pi = static_cast<PointCloud*>(ci_)->points_[ii];
  Point& pj = static_cast<PointCloud*>(cj_)->points_[jj];
  Eigen::Matrix3d Ri, Rj;
  for (int r = 0; r < 3; r++)
    for (int q = 0; q < 3; q++) {
      Ri(r, q) = rot_t_i[3 * r + q];
      Rj(r, q) = rot_t_j[3 * r + q];
    }
  Eigen::Vector3d ppi(pi.p_[0], pi.p_[1], pi.p_[2]);
  Eigen::Vector3d ppj(pj.p_[0], pj.p_[1], pj.p_[2]);
  Eigen::Vector3d npi(pi.n_[0], pi.n_[1], pi.n_[2]);
  Eigen::Vector3d diff = ppi - ppj;
  double b = diff.dot(npi);
  for (int q = 0; q < 6; q++) {
    idx[q] = i * 6 + q;
    idx[6 + q] = j * 6 + q;
  }
  Eigen::Vector3d temp = ppj.cross(npi);
  val[0] = temp(0); val[1] = temp(1); val[2] = temp(2);
  val[3] = npi(0); val[4] = npi(1); val[5] = npi(2);
  val[6] = -temp(0); val[7] = -temp(1); val[8] = -temp(2);
  val[9] = -npi(0); val[10] = -npi(1); val[11] = -npi(2);
  Eigen::Vector3d dTi = Ri * npi;
  Eigen::Vector3d dTj = -Rj * npi;
  for (int ll = 0; ll < 8; ll++)
    for (int xyz = 0; xyz < 3; xyz++) {
      idx[12 + ll * 3 + xyz] = 6 * num_ + pi.idx_[ll] + xyz;
      val[12 + ll * 3 + xyz] = pi.val_[ll] * dTi(xyz);
    }
  for (int ll = 0; ll < 8; ll++)
    for (int xyz = 0; xyz < 3; xyz++) {
      idx[12 + 24 + ll * 3 + xyz] = 6 * num_ + pj.idx_[ll] + xyz;
      val[12 + 24 + ll * 3 + xyz] = pj.val_[ll] * dTj(xyz);
    }
  *b_out = b;
}

}  // extern "C"
