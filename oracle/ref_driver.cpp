// ref_driver.cpp -- TEST INFRASTRUCTURE ONLY (oracle).
//
// Thin C-ABI driver around the UNMODIFIED reference classes CIntegrateApp / TSDFVolume /
// ControlGrid (compiled in place from /root/reference/Integrate/*.cpp by oracle/Makefile into
// oracle/_ref/libref_tsdf.so).  It lets the tests and bench.py's cpu_baseline leg drive the
// reference per frame (CIntegrateApp::Execute, IntegrateApp.cpp:190-226 -> Reproject :228-269,
// TSDFVolume::ScaleDepth TSDFVolume.cpp:19-36, TSDFVolume::Integrate :38-67) and read back the
// hashed volume units (TSDFVolume.h:27) voxel for voxel -- the PCD written by SaveWorld drops
// weight_ and filters voxels (TSDFVolume.cpp:118), so it cannot serve as a voxel-level oracle.
//
// No reference arithmetic is restated here; every number comes out of the reference's own code.
// The product library never links or loads this file.
#include "StdAfx.h"
// Pull in every system / Eigen header the reference headers use BEFORE the access hack below, so the
// hack only ever sees the reference's own class definitions.
#include <vector>
#include <fstream>
#include <sstream>
#include <unordered_set>
#include <unordered_map>
#include <algorithm>
#include <Eigen/Dense>
#define private public   // reach CIntegrateApp::Execute / Reproject (private in IntegrateApp.h:88-89)
#include "IntegrateApp.h"
#undef private

#include <algorithm>

namespace {
struct NullGrabber : public pcl::Grabber {};
struct RefApp {
  NullGrabber grabber;
  CIntegrateApp app;
  RefApp() : grabber(), app(grabber, false) {}
};
}  // namespace

extern "C" {

void* ref_app_create() { return new RefApp(); }
void ref_app_destroy(void* h) { delete static_cast<RefApp*>(h); }

// Mirrors Integrate.cpp:64-78: set the public option members, then CIntegrateApp::Init().
// Empty strings leave the defaults.  Returns the trajectory length after Init.
int ref_app_init(void* h, const char* ref_traj, const char* pose_traj, const char* seg_traj, const char* ctr,
                 const char* camera, int num, int resolution, double length, int interval) {
  CIntegrateApp& a = static_cast<RefApp*>(h)->app;
  a.traj_filename_ = ref_traj ? ref_traj : "";
  a.pose_filename_ = pose_traj ? pose_traj : "";
  a.seg_filename_ = seg_traj ? seg_traj : "";
  a.ctr_filename_ = ctr ? ctr : "";
  a.camera_filename_ = camera ? camera : "";
  a.ctr_num_ = num;
  a.ctr_resolution_ = resolution;
  a.ctr_length_ = length;
  a.ctr_interval_ = interval;
  a.Init();
  return (int)a.traj_.data_.size();
}

void ref_app_set_window(void* h, int start_from, int end_at) {
  CIntegrateApp& a = static_cast<RefApp*>(h)->app;
  a.start_from_ = start_from;
  a.end_at_ = end_at;
}

int ref_app_ctr_num(void* h) { return static_cast<RefApp*>(h)->app.ctr_num_; }

// One grabber delivery + one main-loop turn: what source_cb2_trigger (IntegrateApp.cpp:170-188) and
// Execute(true) do for frame `frame_id` (1-based).  depth_inout receives depth_ after the call
// (i.e. after Reproject when a control grid is active).  Returns exit_.
int ref_app_execute(void* h, int frame_id, unsigned short* depth_inout, float* scaled_out) {
  CIntegrateApp& a = static_cast<RefApp*>(h)->app;
  const size_t n = (size_t)a.cols_ * a.rows_;
  memcpy(&a.depth_[0], depth_inout, n * sizeof(unsigned short));
  a.frame_id_ = frame_id;
  a.Execute(true);
  memcpy(depth_inout, &a.depth_[0], n * sizeof(unsigned short));
  if (scaled_out) memcpy(scaled_out, &a.scaled_depth_[0], n * sizeof(float));
  return a.exit_ ? 1 : 0;
}

// Direct access to the numeric core for unit tests (no gating logic).
void ref_scale_depth(void* h, const unsigned short* depth, float* scaled) {
  CIntegrateApp& a = static_cast<RefApp*>(h)->app;
  const size_t n = (size_t)a.cols_ * a.rows_;
  std::vector<unsigned short> d(depth, depth + n);
  std::vector<float> s(n);
  a.volume_.ScaleDepth(d, s);
  memcpy(scaled, s.data(), n * sizeof(float));
}

void ref_integrate(void* h, const unsigned short* depth, const double* T_rowmajor16) {
  CIntegrateApp& a = static_cast<RefApp*>(h)->app;
  const size_t n = (size_t)a.cols_ * a.rows_;
  std::vector<unsigned short> d(depth, depth + n);
  std::vector<float> s(n);
  Eigen::Matrix4d T;
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) T(r, c) = T_rowmajor16[r * 4 + c];
  a.volume_.ScaleDepth(d, s);
  a.volume_.Integrate(d, s, T);
}

void ref_set_camera(void* h, const float cam6[6]) {
  CameraParam& c = static_cast<RefApp*>(h)->app.volume_.camera_;
  c.fx_ = cam6[0]; c.fy_ = cam6[1]; c.cx_ = cam6[2]; c.cy_ = cam6[3];
  c.ICP_trunc_ = cam6[4]; c.integration_trunc_ = cam6[5];
}

int ref_unit_count(void* h) { return (int)static_cast<RefApp*>(h)->app.volume_.data_.size(); }

// Keys in ascending order (the reference's unordered_map order is not canonical).
void ref_unit_keys(void* h, int* keys) {
  TSDFVolume& v = static_cast<RefApp*>(h)->app.volume_;
  int n = 0;
  for (std::unordered_map<int, TSDFVolumeUnit::Ptr>::iterator it = v.data_.begin(); it != v.data_.end(); ++it)
    keys[n++] = it->first;
  std::sort(keys, keys + n);
}

int ref_read_unit(void* h, int key, float* sdf, float* weight) {
  TSDFVolume& v = static_cast<RefApp*>(h)->app.volume_;
  std::unordered_map<int, TSDFVolumeUnit::Ptr>::iterator it = v.data_.find(key);
  if (it == v.data_.end()) return -1;
  const size_t n = 64 * 64 * 64;
  if (sdf) memcpy(sdf, it->second->sdf_, n * sizeof(float));
  if (weight) memcpy(weight, it->second->weight_, n * sizeof(float));
  return 0;
}

void ref_save_world(void* h, const char* filename) { static_cast<RefApp*>(h)->app.volume_.SaveWorld(filename); }

}  // extern "C"
