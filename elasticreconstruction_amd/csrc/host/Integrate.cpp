// Integrate -- drop-in for the reference's Integrate.exe (Integrate/Integrate.cpp:12-88,
// Integrate/IntegrateApp.cpp:43-226) with the numeric core on MI355X through liber_hip.so.
//
// Every reference flag keeps its name, default and meaning:
//   --ref_traj --pose_traj --seg_traj --ctr --num --resolution --length --interval --camera --save_to
//   --start_from --end_at  -oni <file>
// and the same files are read/written (.log / .ctr / camera file in, world.pcd out).  The reference can
// only read OpenNI devices and .oni recordings (Integrate.cpp:46-60); neither can exist on this machine,
// so the depth stream comes from one of (additive flags):
//   -oni <file> | --depth_raw <file>   raw stream of 640x480 little-endian uint16 frames, FrameID = 1,2,...
//                                      (-oni with a real OpenNI recording is rejected with a clear message)
//   --depth_list <txt>                 one 16-bit grayscale PNG path per line, line i = frame i
// New, additive: --device <gpu> (0), --max_units <n> (2048), --batch <frames fused per launch> (64).
//
// Control flow = CIntegrateApp::StartMainLoop/Execute: 1-based frame ids, frame_ == -1 skips a frame,
// start_from/end_at window, the end-of-trajectory off-by-one (frame N of an N-entry trajectory is never
// integrated, IntegrateApp.cpp:200-203), Reproject's frame_id > interval*num exit (:230-233).  Frames that
// pass the gates are queued and flushed to the GPU in batches; the device applies a batch in frame order
// per voxel, so the volume equals frame-by-frame execution bit for bit.
#include <chrono>
#include <cstdio>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "../er_mat4.h"
#include "er_formats.h"
#include "er_hip.h"

using erfmt::FramedTransformation;

namespace {

int print_help() {
  std::cout << "\nApplication parameters:" << std::endl;
  std::cout << "    --help, -h                      : print this message" << std::endl;
  std::cout << "    --ref_traj <log_file>           : use a reference trajectory file" << std::endl;
  std::cout << "    --pose_traj <log_file>          : use a pose trajectory file to create a reference trajectory" << std::endl;
  std::cout << "    --seg_traj <log_file>           : trajectory within each fragment - must have" << std::endl;
  std::cout << "    --ctr <ctr_file>                : enables distortion, must specify the following parameters" << std::endl;
  std::cout << "    --num <number>                  : number of pieces, important parameter" << std::endl;
  std::cout << "    --resolution <resolution>       : default - 8" << std::endl;
  std::cout << "    --length <length>               : default - 3.0" << std::endl;
  std::cout << "    --interval <interval>           : default - 50" << std::endl;
  std::cout << "    --camera <param_file>           : load camera parameters" << std::endl;
  std::cout << "    --save_to <pcd_file>            : output file, default - world.pcd" << std::endl;
  std::cout << "    --start_from <frame_id>         : frames before frame_id will be skipped" << std::endl;
  std::cout << "    --end_at <frame_id>             : frames after frame_id will be skipped" << std::endl;
  std::cout << "Valid depth data sources:" << std::endl;
  std::cout << "    -oni <raw_file> | --depth_raw <raw_file> : 640x480 uint16 frames; --depth_list <txt> : 16-bit PNG per line" << std::endl;
  std::cout << "MI355X options:" << std::endl;
  std::cout << "    --device <gpu> (0)  --max_units <n> (2048)  --batch <frames> (64)" << std::endl;
  return 0;
}

struct DepthSource {
  virtual ~DepthSource() {}
  virtual bool next(std::vector<uint16_t>& frame, int& frame_id) = 0;
};

struct RawStream : DepthSource {
  FILE* f = nullptr;
  int id = 0;
  size_t px;
  RawStream(const std::string& p, size_t pixels) : px(pixels) { f = fopen(p.c_str(), "rb"); }
  ~RawStream() { if (f) fclose(f); }
  bool next(std::vector<uint16_t>& frame, int& frame_id) override {
    frame.resize(px);
    if (!f || fread(frame.data(), sizeof(uint16_t), px, f) != px) return false;
    frame_id = ++id;
    return true;
  }
};

struct PngList : DepthSource {
  std::vector<std::string> files;
  size_t at = 0;
  int cols, rows;
  PngList(const std::string& list, int c, int r) : cols(c), rows(r) {
    FILE* f = fopen(list.c_str(), "r");
    if (!f) return;
    std::string dir;
    size_t cut = list.find_last_of("/\\");
    if (cut != std::string::npos) dir = list.substr(0, cut + 1);
    char buf[4096];
    while (fgets(buf, sizeof buf, f)) {
      std::string s(buf);
      while (!s.empty() && (s.back() == '\n' || s.back() == '\r' || s.back() == ' ')) s.pop_back();
      if (s.empty() || s[0] == '#') continue;
      files.push_back((s[0] == '/' || dir.empty()) ? s : dir + s);
    }
    fclose(f);
  }
  bool next(std::vector<uint16_t>& frame, int& frame_id) override {
    if (at >= files.size()) return false;
    int w = 0, h = 0;
    if (!erfmt::load_png16(files[at], w, h, frame) || w != cols || h != rows) {
      fprintf(stderr, "Cannot read %dx%d 16-bit depth PNG %s\n", cols, rows, files[at].c_str());
      return false;
    }
    frame_id = (int)++at;
    return true;
  }
};

struct App {
  // CIntegrateApp members (IntegrateApp.h:36-77), same names
  int cols_ = 640, rows_ = 480;
  bool exit_ = false;
  int frame_id_ = 0;
  std::vector<FramedTransformation> traj_, seg_traj_, pose_traj_;
  std::string traj_filename_, pose_filename_, seg_filename_, camera_filename_, ctr_filename_, pcd_filename_ = "world.pcd";
  std::vector<float> grids_;
  int ctr_resolution_ = 8, ctr_interval_ = 50, ctr_num_ = 0;
  double ctr_length_ = 3.0;
  int start_from_ = -1, end_at_ = 100000000;
  // device side
  er_tsdf_t volume_ = nullptr;
  int device_ = 0, max_units_ = 2048, batch_ = ER_MAX_BATCH;
  std::vector<uint16_t> q_depth_;
  std::vector<double> q_T_, q_seg_, q_madj_;
  std::vector<int> q_gi_;
  long frames_integrated_ = 0;

  bool Init() {                                                        // IntegrateApp.cpp:43-79
    float cam[6];
    erfmt::load_camera(erfmt::file_exists(camera_filename_) ? camera_filename_ : std::string(), cam);
    if (er_tsdf_create(cols_, rows_, cam, max_units_, device_, &volume_) != 0) {
      fprintf(stderr, "Integrate: %s\n", er_last_error());
      return false;
    }
    if (ctr_num_ > 0 && erfmt::file_exists(ctr_filename_) && erfmt::file_exists(seg_filename_)) {
      erfmt::load_ctr(ctr_filename_, ctr_num_, ctr_resolution_, grids_);
    } else {
      ctr_num_ = 0;
    }
    if (erfmt::file_exists(traj_filename_)) erfmt::load_log(traj_filename_, traj_);
    if (erfmt::file_exists(seg_filename_)) {
      erfmt::load_log(seg_filename_, seg_traj_);
      if (erfmt::file_exists(pose_filename_)) {
        erfmt::load_log(pose_filename_, pose_traj_);
        traj_.clear();
        for (int i = 0; i < (int)pose_traj_.size(); i++)
          for (int j = 0; j < ctr_interval_; j++) {
            const int idx = i * ctr_interval_ + j;
            if (idx >= (int)seg_traj_.size()) { fprintf(stderr, "Integrate: --seg_traj has fewer than %d entries\n", idx + 1); return false; }
            FramedTransformation t;
            t.id1 = idx; t.id2 = idx; t.frame = idx + 1;
            er::mat4_mul(pose_traj_[i].T, seg_traj_[idx].T, t.T);     // :71
            traj_.push_back(t);
          }
        printf("Trajectory created from pose and segment trajectories.\n");
      }
    }
    return true;
  }

  bool Flush() {
    const int n = (int)q_gi_.size();
    if (n == 0) return true;
    er_warp w;
    const er_warp* wp = nullptr;
    if (ctr_num_ > 0) {
      w.ctr = grids_.data();
      w.num_grids = ctr_num_;
      w.resolution = ctr_resolution_;
      w.length = (float)ctr_length_;                                   // ControlGrid::Load( f, res, float len )
      w.grid_index = q_gi_.data();
      w.seg = q_seg_.data();
      w.madj = q_madj_.data();
      wp = &w;
    }
    if (er_tsdf_integrate_frames(volume_, n, q_depth_.data(), 0, q_T_.data(), wp) != 0) {
      fprintf(stderr, "Integrate: %s\n", er_last_error());
      return false;
    }
    frames_integrated_ += n;
    q_depth_.clear(); q_T_.clear(); q_seg_.clear(); q_madj_.clear(); q_gi_.clear();
    return CheckStatus(false);
  }

  // The reference grows data_ on demand; here the unit pool is fixed (--max_units).  Poll the device's sticky flags after
  // every batch so that an exhausted pool stops the run at once instead of surfacing in SaveWorld after the whole sequence.
  bool CheckStatus(bool final_report) {
    int flags = 0;
    long oor = 0;
    if (er_tsdf_status(volume_, &flags, &oor) != 0) { fprintf(stderr, "Integrate: %s\n", er_last_error()); return false; }
    if (flags & ER_STATUS_POOL_EXHAUSTED) {
      fprintf(stderr, "Integrate: the volume needs more than --max_units %d units (2 MiB each); re-run with a larger --max_units\n", max_units_);
      return false;
    }
    if (flags & ER_STATUS_TABLE_FULL) { fprintf(stderr, "Integrate: unit hash table full (raise --max_units)\n"); return false; }
    if (final_report && oor > 0)
      fprintf(stderr, "Integrate: warning: %ld depth pixels fell outside the 512^3-unit index range (beyond +-96 m) and were skipped\n", oor);
    return true;
  }

  bool Execute(const std::vector<uint16_t>& depth) {                   // IntegrateApp.cpp:190-226
    if (frame_id_ - 1 < 0 || frame_id_ - 1 >= (int)traj_.size()) {    // the reference reads out of bounds here
      exit_ = true;
      return true;
    }
    if (traj_[frame_id_ - 1].frame == -1) return true;
    if (frame_id_ >= (int)traj_.size()) { exit_ = true; return true; }
    if (frame_id_ % 100 == 0) printf("Frames processed : %d / %d\n", frame_id_, (int)traj_.size());
    if (frame_id_ < start_from_ || frame_id_ > end_at_) {
      if (frame_id_ > end_at_) { printf("Reaching the specified end point.\n"); exit_ = true; }
      return true;
    }
    const double* T = traj_[frame_id_ - 1].T;
    if (ctr_num_ > 0) {                                                // Reproject, :228-243
      if (frame_id_ > ctr_interval_ * ctr_num_) { exit_ = true; return true; }
      if (seg_traj_.empty() || frame_id_ - 1 >= (int)seg_traj_.size()) {  // the reference indexes seg_traj_ unchecked (:243,:251)
        fprintf(stderr, "Integrate: --seg_traj has %d entries, frame %d needs entry %d\n", (int)seg_traj_.size(), frame_id_, frame_id_ - 1);
        return false;
      }
      const int chunk = (frame_id_ - 1) / ctr_interval_;
      double Tinv[16], S0inv[16], tmp[16], madj[16];
      if (!er::mat4_inverse(T, Tinv) || !er::mat4_inverse(seg_traj_[0].T, S0inv)) { fprintf(stderr, "Integrate: singular pose\n"); return false; }
      er::mat4_mul(Tinv, traj_[0].T, tmp);
      er::mat4_mul(tmp, S0inv, madj);                                  // TiT0Ai_adj, :243
      q_madj_.insert(q_madj_.end(), madj, madj + 16);
      q_seg_.insert(q_seg_.end(), seg_traj_[frame_id_ - 1].T, seg_traj_[frame_id_ - 1].T + 16);
      q_gi_.push_back(chunk);
    } else {
      q_gi_.push_back(0);
    }
    q_depth_.insert(q_depth_.end(), depth.begin(), depth.end());
    q_T_.insert(q_T_.end(), T, T + 16);
    if ((int)q_gi_.size() >= batch_) return Flush();
    return true;
  }

  bool SaveWorld() {                                                   // TSDFVolume::SaveWorld, TSDFVolume.cpp:104-132
    long n = 0;
    if (er_tsdf_extract_world(volume_, nullptr, 0, &n) != 0) { fprintf(stderr, "Integrate: %s\n", er_last_error()); return false; }
    std::vector<float> pts((size_t)n * 4);
    if (n > 0 && er_tsdf_extract_world(volume_, pts.data(), n, &n) != 0) { fprintf(stderr, "Integrate: %s\n", er_last_error()); return false; }
    if (!erfmt::save_pcd_xyzi(pcd_filename_, pts.data(), (size_t)n)) { fprintf(stderr, "Integrate: cannot write %s\n", pcd_filename_.c_str()); return false; }
    printf("%ld voxel points have been written.\n", n);
    return true;
  }
};

}  // namespace

int main(int argc, char* argv[]) {
  using namespace erfmt;
  if (argc == 1 || find_switch(argc, argv, "--help") || find_switch(argc, argv, "-h")) return print_help();

  App app;
  std::string raw_file, list_file, dev_name;
  if (parse_argument(argc, argv, "-dev", dev_name) > 0) {
    std::cout << "Can't open depth source" << std::endl;               // no OpenNI device on this platform
    return -1;
  }
  parse_argument(argc, argv, "-oni", raw_file);
  parse_argument(argc, argv, "--depth_raw", raw_file);
  parse_argument(argc, argv, "--depth_list", list_file);
  if (raw_file.size() > 4 && raw_file.substr(raw_file.size() - 4) == ".oni") {
    std::cout << "Can't open depth source (OpenNI .oni recordings are not supported here; convert the depth stream to "
                 "raw uint16 frames (--depth_raw) or 16-bit PNGs (--depth_list))" << std::endl;
    return -1;
  }
  std::unique_ptr<DepthSource> source;
  if (!list_file.empty()) {
    source.reset(new PngList(list_file, app.cols_, app.rows_));
  } else if (!raw_file.empty() && file_exists(raw_file)) {
    source.reset(new RawStream(raw_file, (size_t)app.cols_ * app.rows_));
  } else {
    std::cout << "Can't open depth source" << std::endl;
    return -1;
  }

  parse_argument(argc, argv, "--ref_traj", app.traj_filename_);
  parse_argument(argc, argv, "--pose_traj", app.pose_filename_);
  parse_argument(argc, argv, "--seg_traj", app.seg_filename_);
  parse_argument(argc, argv, "--camera", app.camera_filename_);
  parse_argument(argc, argv, "--save_to", app.pcd_filename_);
  parse_argument(argc, argv, "--start_from", app.start_from_);
  parse_argument(argc, argv, "--end_at", app.end_at_);
  parse_argument(argc, argv, "--ctr", app.ctr_filename_);
  parse_argument(argc, argv, "--num", app.ctr_num_);
  parse_argument(argc, argv, "--resolution", app.ctr_resolution_);
  parse_argument(argc, argv, "--length", app.ctr_length_);
  parse_argument(argc, argv, "--interval", app.ctr_interval_);
  parse_argument(argc, argv, "--device", app.device_);
  parse_argument(argc, argv, "--max_units", app.max_units_);
  parse_argument(argc, argv, "--batch", app.batch_);
  if (app.batch_ < 1) app.batch_ = 1;
  if (app.batch_ > ER_MAX_BATCH) app.batch_ = ER_MAX_BATCH;

  if (!app.Init()) return 1;
  int rc = 0;
  {
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint16_t> frame;
    while (!app.exit_) {
      int id = 0;
      if (!source->next(frame, id)) break;                             // end of stream (reference: ten timeouts, :125)
      app.frame_id_ = id;
      if (!app.Execute(frame)) { rc = 1; break; }
    }
    if (rc == 0 && !app.Flush()) rc = 1;
    if (rc == 0 && (er_tsdf_synchronize(app.volume_) != 0 || !app.CheckStatus(true))) rc = 1;
    if (rc == 0 && !app.SaveWorld()) rc = 1;
    std::cout << "Total " << app.frame_id_ << " frames processed." << std::endl;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::cerr << "Integrate All took " << ms << "ms." << std::endl;
    if (ms > 0 && app.frames_integrated_ > 0)
      std::cerr << app.frames_integrated_ << " frames integrated, " << 1000.0 * app.frames_integrated_ / ms << " frames/s end to end (incl. file I/O)" << std::endl;
  }
  er_tsdf_destroy(app.volume_);
  return rc;
}
