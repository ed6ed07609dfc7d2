// ref_corres_driver.cpp -- TEST INFRASTRUCTURE.  C-ABI over the reference's own BuildCorrespondence/CorresApp.{h,cpp}
// (compiled IN PLACE, unmodified, against oracle/stub_corres) so that tests and bench.py's cpu_baseline leg can run
// CCorresApp::Registration / FindCorrespondence on in-memory clouds and read the results in full precision (the program
// rounds to 8 decimals, Helper.h:46-50).  Nothing is restated here: every rcorres_* call forwards to a reference method or
// reads a reference member.  Built into oracle/_ref/libref_corres.so (as written: the reference's num_threads( 8 )) and
// oracle/_ref/libref_corres_uncapped.so (the clause erased with -D'num_threads(x)=').
//
// rcorres_icp is the one exception: it runs the STUB's restatement of pcl::IterativeClosestPoint exactly as
// CorresApp.cpp:295-306 configures it and also returns the iteration count -- the independent second statement of the PCL 1.7
// loop (kd-tree + the reference's vendored Eigen) that oracle/icp_oracle.cpp (uniform grid + hand-written LU) is checked against.
#include "StdAfx.h"
#include "CorresApp.h"
#include <pcl/registration/icp.h>
#include <pcl/registration/transformation_estimation_point_to_plane_lls.h>

typedef pcl::PointCloud<pcl::PointXYZRGBNormal> CloudT;

static CloudT::Ptr make_cloud(const float* xyz, const float* nrm, int n) {
  CloudT::Ptr c(new CloudT);
  c->points.resize((size_t)n);
  for (int k = 0; k < n; k++) {
    pcl::PointXYZRGBNormal& p = c->points[k];
    p.x = xyz[3 * k]; p.y = xyz[3 * k + 1]; p.z = xyz[3 * k + 2];
    p.normal_x = nrm[3 * k]; p.normal_y = nrm[3 * k + 1]; p.normal_z = nrm[3 * k + 2];
  }
  c->width = (unsigned)n; c->height = 1;
  return c;
}

extern "C" {

void* rcorres_create(const char* out_dir_with_slash) {
  CCorresApp* a = new CCorresApp();
  if (er_stub::quiet()) std::cout.setstate(std::ios_base::failbit);   // Registration prints every matrix through cout (CorresApp.cpp:309-311)
  memset(a->m_pDirName, 0, 1024);
  if (out_dir_with_slash) strncpy(a->m_pDirName, out_dir_with_slash, 1023);
  else a->save_corres_ = false;
  return a;
}
void rcorres_destroy(void* h) { delete static_cast<CCorresApp*>(h); }

// the values BuildCorrespondence.cpp:52-60 sets from the command line (reg_dist also sets dist_thresh_ = reg_dist / 2)
void rcorres_set_params(void* h, double reg_dist, double dist_thresh, double reg_ratio, int reg_num, int output_information) {
  CCorresApp& a = *static_cast<CCorresApp*>(h);
  a.reg_dist_ = reg_dist; a.dist_thresh_ = dist_thresh; a.reg_ratio_ = reg_ratio; a.reg_num_ = reg_num;
  a.output_information_ = output_information != 0;
}
int rcorres_add_cloud(void* h, const float* xyz, const float* nrm, int n) {     // pointclouds_[ i ] after LoadData's NaN filter
  CCorresApp& a = *static_cast<CCorresApp*>(h);
  a.pointclouds_.push_back(make_cloud(xyz, nrm, n));
  a.num_ = (int)a.pointclouds_.size();
  return a.num_ - 1;
}
void rcorres_add_pair(void* h, int id1, int id2, int frame, const double* T16_rowmajor) {
  CCorresApp& a = *static_cast<CCorresApp*>(h);
  Eigen::Matrix4d T;
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T(r, c) = T16_rowmajor[4 * r + c];
  a.corres_traj_.data_.push_back(FramedTransformation(id1, id2, frame, T));
}
void rcorres_blacklist(void* h, int id) { static_cast<CCorresApp*>(h)->blacklist_.insert(id); }
void rcorres_registration(void* h) { static_cast<CCorresApp*>(h)->Registration(); }
void rcorres_find_correspondence(void* h) { static_cast<CCorresApp*>(h)->FindCorrespondence(); }
int rcorres_num_pairs(void* h) { return (int)static_cast<CCorresApp*>(h)->corres_traj_.data_.size(); }
void rcorres_get_pair(void* h, int k, int* ids3, double* T16_rowmajor, double* info36_rowmajor) {
  CCorresApp& a = *static_cast<CCorresApp*>(h);
  const FramedTransformation& t = a.corres_traj_.data_[k];
  ids3[0] = t.id1_; ids3[1] = t.id2_; ids3[2] = t.frame_;
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T16_rowmajor[4 * r + c] = t.transformation_(r, c);
  if (info36_rowmajor && k < (int)a.corres_info_.data_.size())
    for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) info36_rowmajor[6 * r + c] = a.corres_info_.data_[k].information_(r, c);
}
double rcorres_overlap_ratio(void* h, double length, const double* T16_rowmajor) {   // CorresApp.h:64-81
  CCorresApp& a = *static_cast<CCorresApp*>(h);
  a.length_ = length;
  Eigen::Matrix4d T;
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) T(r, c) = T16_rowmajor[4 * r + c];
  return a.GetVolumeOverlapRatio(T);
}

// pcl::IterativeClosestPoint as CorresApp.cpp:295-306 sets it up (stub restatement, see the header of this file).
int rcorres_icp(const float* sxyz, const float* snrm, int sn, const float* txyz, const float* tnrm, int tn, const float* guess16_rowmajor,
                double max_dist, int max_iter, double eps, float* out16_rowmajor, int* iterations, int* converged, double* fitness) {
  CloudT::Ptr pcd1 = make_cloud(sxyz, snrm, sn), pcd0 = make_cloud(txyz, tnrm, tn);
  CloudT transformed;
  pcl::IterativeClosestPoint<pcl::PointXYZRGBNormal, pcl::PointXYZRGBNormal> icp;
  typedef pcl::registration::TransformationEstimationPointToPlaneLLS<pcl::PointXYZRGBNormal, pcl::PointXYZRGBNormal> PointToPlane;
  boost::shared_ptr<PointToPlane> point_to_plane(new PointToPlane);
  icp.setInputCloud(pcd1);
  icp.setInputTarget(pcd0);
  icp.setMaxCorrespondenceDistance(max_dist);
  icp.setMaximumIterations(max_iter);
  icp.setTransformationEpsilon(eps);
  icp.setTransformationEstimation(point_to_plane);
  Eigen::Matrix4f g;
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) g(r, c) = guess16_rowmajor[4 * r + c];
  icp.align(transformed, g);
  Eigen::Matrix4f f = icp.getFinalTransformation();
  for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) out16_rowmajor[4 * r + c] = f(r, c);
  if (iterations) *iterations = icp.iterations();
  if (converged) *converged = icp.hasConverged() ? 1 : 0;
  if (fitness) *fitness = icp.getFitnessScore();
  return 0;
}

}  // extern "C"
