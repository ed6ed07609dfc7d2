// ref_ransac_driver.cpp -- TEST INFRASTRUCTURE.  C-ABI over the reference's own GlobalRegistration/RansacCurvature.h (included
// IN PLACE, unmodified, against oracle/stub_corres): getFitness (:661-704), getInformation (:707-733) and align_redux (:751-817)
// run as REFERENCE code on in-memory clouds; pins oracle/icp_oracle.cpp's icp_ransac_fitness / icp_ransac_inliers.
// PCL surface used by those three methods (stub): transformPointCloud( Matrix4f ), KdTreeFLANN::nearestKSearch (exact kd-tree).
//
// The header was written for MSVC, which looks names up at instantiation time; align_redux uses the inherited members
// `indices_`, `previous_transformation_` and calls `initCompute()` / `deinitCompute()` WITHOUT a using-declaration or `this->`
// (RansacCurvature.h:754-757,779,816).  Under two-phase lookup those are non-dependent names and must resolve where the template
// is defined, so this file declares namespace-scope objects of those names BEFORE including the header: align_redux then reads
// `indices_` (all points, what pcl::PCLBase::initCompute leaves when no indices were set) from here.  The reference source
// itself is untouched.
#include "er_corres_stub.h"

static pcl::IndicesPtr indices_(new std::vector<int>);
static Eigen::Matrix4f previous_transformation_ = Eigen::Matrix4f::Identity();
static bool initCompute() { return true; }
static void deinitCompute() {}

#include "RansacCurvature.h"

typedef RansacCurvature<pcl::PointNormal, pcl::PointNormal, pcl::FPFHSignature33> RansacT;
struct Probe : public RansacT {
  void fitness(std::vector<int>& a, std::vector<int>& b, float& f) { getFitness(a, b, f); }
  void set_final(const Eigen::Matrix4f& m) { final_transformation_ = m; }
  const std::vector<int>& inliers_target() const { return inliers_target_; }
};

static pcl::PointCloud<pcl::PointNormal>::Ptr make_cloud(const float* xyz, const float* nrm, int n) {
  pcl::PointCloud<pcl::PointNormal>::Ptr c(new pcl::PointCloud<pcl::PointNormal>);
  c->points.resize((size_t)n);
  for (int k = 0; k < n; k++) {
    pcl::PointNormal& p = c->points[k];
    p.x = xyz[3 * k]; p.y = xyz[3 * k + 1]; p.z = xyz[3 * k + 2];
    p.normal_x = nrm[3 * k]; p.normal_y = nrm[3 * k + 1]; p.normal_z = nrm[3 * k + 2];
  }
  c->width = (unsigned)n; c->height = 1;
  return c;
}

extern "C" {

void* rransac_create(const float* sxyz, const float* snrm, int sn, const float* txyz, const float* tnrm, int tn, double max_corr_dist,
                     float inlier_fraction, int inlier_number) {
  Probe* p = new Probe();
  p->setInputCloud(make_cloud(sxyz, snrm, sn));                 // GlobalRegistration.cpp:303-313
  p->setInputTarget(make_cloud(txyz, tnrm, tn));
  p->setMaxCorrespondenceDistance(max_corr_dist);
  p->setInlierFraction(inlier_fraction);
  p->setInlierNumber(inlier_number);
  return p;
}
void rransac_destroy(void* h) { delete static_cast<Probe*>(h); }

// getFitness for final_transformation_ = M: inlier lists in point order and the float32 fitness the reference accumulates.
int rransac_fitness(void* h, const float* M16_rowmajor, int* inliers, int* inliers_target, float* fitness) {
  Probe& p = *static_cast<Probe*>(h);
  Eigen::Matrix4f M;
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) M(r, c) = M16_rowmajor[4 * r + c];
  p.set_final(M);
  std::vector<int> a, b;
  float f = 0.f;
  p.fitness(a, b, f);
  if (inliers) memcpy(inliers, a.data(), a.size() * sizeof(int));
  if (inliers_target) memcpy(inliers_target, b.data(), b.size() * sizeof(int));
  if (fitness) *fitness = f;
  return (int)a.size();
}

// align_redux( output, guess ) + getInformation(), the calls of GlobalRegistration.cpp:318-322.  Returns hasConverged().
int rransac_align_redux(void* h, const float* guess16_rowmajor, int n_source, int* n_inliers, int* inliers, int* inliers_target,
                        double* info_source36_rowmajor, double* info_target36_rowmajor) {
  Probe& p = *static_cast<Probe*>(h);
  indices_->resize((size_t)n_source);
  for (int i = 0; i < n_source; i++) (*indices_)[i] = i;
  Eigen::Matrix4f g;
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) g(r, c) = guess16_rowmajor[4 * r + c];
  pcl::PointCloud<pcl::PointNormal> out;
  p.align_redux(out, g);
  p.getInformation();
  const std::vector<int>& a = p.getInliers();
  const std::vector<int>& b = p.inliers_target();
  *n_inliers = (int)a.size();
  if (inliers) memcpy(inliers, a.data(), a.size() * sizeof(int));
  if (inliers_target) memcpy(inliers_target, b.data(), b.size() * sizeof(int));
  for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) {
    info_source36_rowmajor[6 * r + c] = p.information_source_(r, c);
    info_target36_rowmajor[6 * r + c] = p.information_target_(r, c);
  }
  return p.hasConverged() ? 1 : 0;
}

}  // extern "C"
