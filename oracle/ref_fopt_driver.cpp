// ref_fopt_driver.cpp -- C-ABI over the reference's own FragmentOptimizer/PointCloud.{h,cpp} (compiled IN PLACE, unmodified,
// against the reference's vendored Eigen): nothing is restated here, every function forwards to a reference method.
// Built into oracle/_ref/libref_fopt.so only where /root/reference exists; it pins the float32 point state of
// oracle/fopt_oracle.cpp bit for bit (tests/test_fopt_oracle.py).  The assembly loops of OptApp.cpp are pinned through the
// whole reference program instead (oracle/_ref/FragmentOptimizer_ref on oracle/cholmod_shim.cpp).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "PointCloud.h"            // the reference's header: Point, PointCloud::GetCoordinate / UpdatePose / UpdateAllPointPN

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

}  // extern "C"
