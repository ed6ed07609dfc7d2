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
//   --depth_list <txt>                 one 16-bit grayscale PNG path per line, line i = frame i (inflated ahead by --decode_threads <n> host threads; default: an eighth of the host's hardware threads, 8..32)
// New, additive: --device <gpu> (0), --max_units <n> (2048), --batch <frames fused per launch> (64),
//   --gpus <N>            N GPUs (devices --device ... --device + N - 1), one host thread per GPU
//   --shard frame|unit    frame (default, what BASELINE.json names): the active frame range is cut into N contiguous blocks,
//                         every GPU integrates its block into a private volume, then er_tsdf_allreduce merges them: every volume
//                         unit goes to the GPU that observed most of it, which adds the other GPUs' band records in rank order
//                         (weights exact, sdf within 1e-5 of the single-GPU run: float32 summation order).  The merged volume stays
//                         DISTRIBUTED -- SaveWorld is per unit (TSDFVolume.cpp:104-132): world.pcd is assembled from the GPUs'
//                         extractions in ascending key order -- unless --merge_root <g> gathers it on GPU g first;
//                         unit: every GPU is fed all frames and owns the units with er_unit_owner(key, N) == its rank
//                         (er_tsdf_set_unit_shard): no collective, world.pcd BIT-identical to the single-GPU run
//   --force_merge         with --gpus 1: run the frame-split merge anyway (exercises the RCCL path on a 1-GPU box)
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
#include <algorithm>
#include <string>
#include <condition_variable>
#include <mutex>
#include <future>
#include <thread>
#include <vector>

#include "../er_mat4.h"
#include <unistd.h>

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
  std::cout << "    --gpus <N> (1)  --shard frame|unit (frame: frame blocks + the merge by unit owner over RCCL; unit: bit-exact, no collective)  --merge_root <g>  --force_merge" << std::endl;
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
  RawStream(const std::string& p, size_t pixels, long first_id = 1) : px(pixels) {
    f = fopen(p.c_str(), "rb");
    if (f && first_id > 1) { fseek(f, (long)((first_id - 1) * (long)(px * sizeof(uint16_t))), SEEK_SET); id = (int)first_id - 1; }
  }
  static long count(const std::string& p, size_t pixels) {
    FILE* g = fopen(p.c_str(), "rb");
    if (!g) return 0;
    fseek(g, 0, SEEK_END);
    const long n = ftell(g) / (long)(pixels * sizeof(uint16_t));
    fclose(g);
    return n;
  }
  ~RawStream() { if (f) fclose(f); }
  bool next(std::vector<uint16_t>& frame, int& frame_id) override {
    frame.resize(px);
    if (!f || fread(frame.data(), sizeof(uint16_t), px, f) != px) return false;
    frame_id = ++id;
    return true;
  }
};

// 16-bit PNG list.  Inflating one 640 x 480 frame takes a host core 3-5 ms -- two orders of magnitude more than the GPU needs to integrate it -- so the
// files are decoded AHEAD by a few host threads (--decode_threads): thread-safe claim of the next file, a ring of finished frames, next()
// hands them out strictly in list order (the frame ids and their order are the reference's, IntegrateApp.cpp:190-226).
struct PngList : DepthSource {
  std::vector<std::string> files;
  size_t at = 0;
  int cols, rows, nthreads;
  struct Slot { std::vector<uint16_t> px; long idx = -1; bool ready = false, ok = false; };
  std::vector<Slot> ring;
  std::vector<std::thread> workers;
  std::mutex m;
  std::condition_variable cv_ready, cv_free;
  size_t next_claim = 0;
  bool stop = false, started = false;
  PngList(const std::string& list, int c, int r, long first_id = 1, int threads = 8) : cols(c), rows(r), nthreads(std::max(1, std::min(threads, 64))) {
    at = first_id > 1 ? (size_t)(first_id - 1) : 0;
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
  ~PngList() override {
    {
      std::lock_guard<std::mutex> lk(m);
      stop = true;
    }
    cv_free.notify_all();
    for (auto& t : workers) t.join();
  }
  void start() {
    started = true;
    next_claim = at;
    ring.resize((size_t)nthreads * 2);
    for (int t = 0; t < nthreads; t++)
      workers.emplace_back([this] {
        std::vector<uint16_t> px;
        for (;;) {
          size_t i;
          {
            std::unique_lock<std::mutex> lk(m);
            cv_free.wait(lk, [&] { return stop || (next_claim < files.size() && ring[next_claim % ring.size()].idx < 0); });
            if (stop || next_claim >= files.size()) return;
            i = next_claim++;
            ring[i % ring.size()].idx = (long)i;               // claimed: not ready yet
            if (next_claim >= files.size()) cv_free.notify_all();   // (the others can leave)
          }
          int w = 0, h = 0;
          const bool ok = erfmt::load_png16(files[i], w, h, px) && w == cols && h == rows;
          {
            std::lock_guard<std::mutex> lk(m);
            Slot& s = ring[i % ring.size()];
            s.px.swap(px);
            s.ok = ok;
            s.ready = true;
          }
          cv_ready.notify_all();
        }
      });
  }
  bool next(std::vector<uint16_t>& frame, int& frame_id) override {
    if (at >= files.size()) return false;
    if (!started) start();
    bool ok;
    {
      std::unique_lock<std::mutex> lk(m);
      Slot& s = ring[at % ring.size()];
      cv_ready.wait(lk, [&] { return s.idx == (long)at && s.ready; });
      frame.swap(s.px);
      ok = s.ok;
      s.idx = -1;
      s.ready = false;
    }
    cv_free.notify_all();
    if (!ok) {
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
  int rank_ = 0, gpus_ = 1;                                            // this worker / number of workers (--gpus)
  bool unit_shard_ = false;                                            // --shard unit
  std::vector<uint16_t> q_depth_;
  std::vector<double> q_T_, q_seg_, q_madj_;
  std::vector<int> q_gi_;
  long frames_integrated_ = 0;

  bool Init() {                                                        // IntegrateApp.cpp:43-79
    float cam[6];
    erfmt::load_camera(erfmt::file_exists(camera_filename_) ? camera_filename_ : std::string(), cam);
    // (the files first: main() brings the HIP runtime up on a side thread meanwhile)
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
        if (rank_ == 0) printf("Trajectory created from pose and segment trajectories.\n");
      }
    }
    if (er_tsdf_create(cols_, rows_, cam, max_units_, device_, &volume_) != 0) {
      fprintf(stderr, "Integrate: %s\n", er_last_error());
      return false;
    }
    if (unit_shard_ && gpus_ > 1 && er_tsdf_set_unit_shard(volume_, rank_, gpus_) != 0) {
      fprintf(stderr, "Integrate: %s\n", er_last_error());
      return false;
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
    if (frame_id_ % 100 == 0 && (rank_ == 0 || !unit_shard_)) printf("Frames processed : %d / %d\n", frame_id_, (int)traj_.size());
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

  bool ExtractWorld(std::vector<float>& pts) {                         // TSDFVolume::SaveWorld's voxel filter, TSDFVolume.cpp:104-132
    long n = 0;
    if (er_tsdf_extract_world(volume_, nullptr, 0, &n) != 0) { fprintf(stderr, "Integrate: %s\n", er_last_error()); return false; }
    pts.assign((size_t)n * 4, 0.f);
    if (n > 0 && er_tsdf_extract_world(volume_, pts.data(), n, &n) != 0) { fprintf(stderr, "Integrate: %s\n", er_last_error()); return false; }
    return true;
  }

  bool SaveWorld() {
    std::vector<float> pts;
    if (!ExtractWorld(pts)) return false;
    if (!erfmt::save_pcd_xyzi(pcd_filename_, pts.data(), pts.size() / 4)) { fprintf(stderr, "Integrate: cannot write %s\n", pcd_filename_.c_str()); return false; }
    printf("%ld voxel points have been written.\n", (long)(pts.size() / 4));
    return true;
  }
};

// hash_key of the unit a SaveWorld point (voxel index coordinates, TSDFVolume.cpp:119-121) belongs to.
int unit_key_of_point(const float* p) {
  const int xi = ((int)p[0] + 256 * 64) / 64, yi = ((int)p[1] + 256 * 64) / 64, zi = ((int)p[2] + 256 * 64) / 64;
  return xi * 512 * 512 + yi * 512 + zi;
}

// --shard unit: every worker holds a disjoint set of units; world.pcd lists the units in ascending key order exactly like
// the single-GPU program, so the per-worker lists (each already ascending) are merged unit by unit.
bool SaveWorldSharded(std::vector<App>& apps, const std::string& filename) {
  struct Seg { int key, worker; size_t begin, end; };
  std::vector<std::vector<float>> pts(apps.size());
  std::vector<Seg> segs;
  for (size_t w = 0; w < apps.size(); w++) {
    if (!apps[w].ExtractWorld(pts[w])) return false;
    const size_t n = pts[w].size() / 4;
    size_t b = 0;
    while (b < n) {
      const int key = unit_key_of_point(&pts[w][b * 4]);
      size_t e = b + 1;
      while (e < n && unit_key_of_point(&pts[w][e * 4]) == key) e++;
      segs.push_back(Seg{key, (int)w, b, e});
      b = e;
    }
  }
  std::sort(segs.begin(), segs.end(), [](const Seg& a, const Seg& b) { return a.key < b.key; });
  std::vector<float> all;
  for (const Seg& sg : segs) all.insert(all.end(), pts[(size_t)sg.worker].begin() + (long)(sg.begin * 4), pts[(size_t)sg.worker].begin() + (long)(sg.end * 4));
  if (!erfmt::save_pcd_xyzi(filename, all.data(), all.size() / 4)) { fprintf(stderr, "Integrate: cannot write %s\n", filename.c_str()); return false; }
  printf("%ld voxel points have been written.\n", (long)(all.size() / 4));
  return true;
}

}  // namespace

int main(int argc, char* argv[]) {
  erfmt::stage_done("process start");
  er_request_hw_queues(8);                                  // before the first HIP call (include/er_hip.h)
  using namespace erfmt;
  if (argc == 1 || find_switch(argc, argv, "--help") || find_switch(argc, argv, "-h")) return print_help();
  // The HIP runtime takes ~0.1 s to come up: it does so on a side thread while this one parses the trajectory and .ctr files.
  std::future<int> hip_up = std::async(std::launch::async, [] { return er_device_count(); });

  App app;
  std::string raw_file, list_file, dev_name, shard = "frame";
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
  // --decode_threads <n>: host threads that inflate the PNGs of --depth_list ahead (600 frames: 148 ms with 8, 85 ms with 32 on a 256-thread host)
  int decode_threads = (int)std::min(32u, std::max(8u, std::thread::hardware_concurrency() / 8));
  parse_argument(argc, argv, "--decode_threads", decode_threads);
  const size_t px = (size_t)app.cols_ * app.rows_;
  long source_frames = 0;
  if (!list_file.empty()) {
    source_frames = (long)PngList(list_file, app.cols_, app.rows_).files.size();
  } else if (!raw_file.empty() && file_exists(raw_file)) {
    source_frames = RawStream::count(raw_file, px);
  } else {
    std::cout << "Can't open depth source" << std::endl;
    return -1;
  }
  auto open_source = [&](long first_id) -> std::unique_ptr<DepthSource> {
    if (!list_file.empty()) return std::unique_ptr<DepthSource>(new PngList(list_file, app.cols_, app.rows_, first_id, decode_threads));
    return std::unique_ptr<DepthSource>(new RawStream(raw_file, px, first_id));
  };

  int gpus = 1;
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
  parse_argument(argc, argv, "--gpus", gpus);
  parse_argument(argc, argv, "--shard", shard);
  const bool force_merge = find_switch(argc, argv, "--force_merge");
  int merge_root = ER_MERGE_DISTRIBUTED;                                // --merge_root <g>: gather the merged volume on worker g before SaveWorld
  parse_argument(argc, argv, "--merge_root", merge_root);
  // all workers share ONE device: lets a 1-GPU box run the N-worker logic.  --shard unit needs no collective; --shard frame merges through the
  // library's loopback communicator (er_comm_create_loopback: the same protocol, device volumes and kernels, device-to-device copies instead of
  // RCCL, which refuses two ranks on one device)
  const bool same_device = find_switch(argc, argv, "--same_device");
  if (app.batch_ < 1) app.batch_ = 1;
  if (app.batch_ > ER_MAX_BATCH) app.batch_ = ER_MAX_BATCH;
  if (gpus < 1) gpus = 1;
  if (shard != "frame" && shard != "unit") { fprintf(stderr, "Integrate: --shard must be frame or unit\n"); return 1; }
  if (merge_root != ER_MERGE_DISTRIBUTED && (merge_root < 0 || merge_root >= gpus)) { fprintf(stderr, "Integrate: --merge_root must name one of the %d workers\n", gpus); return 1; }
  {
    const char* impl = getenv("ER_MERGE_IMPL");                          // round 5's ring protocol cannot leave the result distributed
    if (impl && std::string(impl) == "ring" && merge_root == ER_MERGE_DISTRIBUTED) merge_root = 0;
  }
  const bool unit_shard = shard == "unit";
  if (same_device && !unit_shard && gpus > 16) { fprintf(stderr, "Integrate: --same_device --shard frame takes at most 16 workers\n"); return 1; }
  if (!same_device && gpus > 1 && er_device_count() > 0 && app.device_ + gpus > er_device_count()) {
    fprintf(stderr, "Integrate: --gpus %d from --device %d, but %d HIP devices are visible\n", gpus, app.device_, er_device_count());
    return 1;
  }

  // one worker (= one copy of the application state, one volume, one host thread) per GPU
  std::vector<App> apps((size_t)gpus, app);
  for (int g = 0; g < gpus; g++) {
    apps[(size_t)g].device_ = app.device_ + (same_device ? 0 : g);
    apps[(size_t)g].rank_ = g;
    apps[(size_t)g].gpus_ = gpus;
    apps[(size_t)g].unit_shard_ = unit_shard;
    if (!apps[(size_t)g].Init()) return 1;
  }
  stage_done("Init (logs, .ctr, volume)");
  const bool merge = !unit_shard && (gpus > 1 || force_merge);
  std::vector<er_comm_t> comms((size_t)gpus, nullptr);
  if (merge) {
    std::vector<int> devs((size_t)gpus);
    for (int g = 0; g < gpus; g++) devs[(size_t)g] = app.device_ + g;
    const int crc = same_device ? er_comm_create_loopback(gpus, app.device_, comms.data()) : er_comm_create_local(gpus, devs.data(), comms.data());
    if (crc != 0) { fprintf(stderr, "Integrate: %s\n", er_last_error()); return 1; }
  }

  // The frame ids Execute can let through (IntegrateApp.cpp:200-216,230-233): [first, last].  --shard frame cuts this range
  // into contiguous blocks; --shard unit hands all of it to every worker.
  const App& a0 = apps[0];
  long first = std::max(1L, (long)a0.start_from_), last = std::min(source_frames, (long)a0.traj_.size() - 1);
  last = std::min(last, (long)a0.end_at_);
  if (a0.ctr_num_ > 0) last = std::min(last, (long)a0.ctr_interval_ * a0.ctr_num_);
  const long active = std::max(0L, last - first + 1);

  int rc = 0;
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<int> wrc((size_t)gpus, 0);
  auto worker = [&](int g) {
    App& w = apps[(size_t)g];
    long lo_id = 1, hi_id = source_frames;                             // single GPU / unit shard: the whole stream, like the reference
    if (gpus > 1 && !unit_shard) {
      int lo = 0, hi = 0;
      er_frame_block((int)active, g, gpus, &lo, &hi);
      lo_id = first + lo;
      hi_id = first + hi - 1;
    }
    std::unique_ptr<DepthSource> source = open_source(lo_id);
    std::vector<uint16_t> frame;
    while (!w.exit_) {
      int id = 0;
      if (!source->next(frame, id)) break;                             // end of stream (reference: ten timeouts, :125)
      if (id > hi_id) break;
      w.frame_id_ = id;
      if (!w.Execute(frame)) { wrc[(size_t)g] = 1; break; }
    }
    if (wrc[(size_t)g] == 0 && !w.Flush()) wrc[(size_t)g] = 1;
    if (wrc[(size_t)g] == 0 && (er_tsdf_synchronize(w.volume_) != 0 || !w.CheckStatus(true))) wrc[(size_t)g] = 1;
    // every rank of the communicator must take part in the merge, failed or not: a rank whose unit pool / hash table overflowed
    // reports that through the merge's first collective and ALL ranks return an error together (er_merge_protocol.h) -- nobody hangs
    if (merge) {
      int nu = 0;
      if (er_tsdf_allreduce(w.volume_, comms[(size_t)g], merge_root, &nu) != 0) { fprintf(stderr, "Integrate: %s\n", er_last_error()); wrc[(size_t)g] = 1; }
      else if (g == 0) {
        long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)er_comm_merge_stats(comms[0], st);
        fprintf(stderr, "Integrate: merged %d GPU volumes over %s, %d units in the union (%lld touched by two or more GPUs, %lld by one), %s\n", gpus,
                same_device ? "the loopback transport (one device)" : "RCCL", nu, st[1], st[2],
                merge_root == ER_MERGE_DISTRIBUTED ? "left distributed by unit owner" : ("gathered on GPU " + std::to_string(merge_root)).c_str());
      }
    }
  };
  if (gpus == 1) {
    worker(0);
  } else {
    std::vector<std::thread> th;
    for (int g = 0; g < gpus; g++) th.emplace_back(worker, g);
    for (auto& t : th) t.join();
  }
  stage_done("frames (read, integrate, sync)");
  long frames_integrated = 0;
  int last_id = 0;
  for (int g = 0; g < gpus; g++) {
    rc |= wrc[(size_t)g];
    frames_integrated += unit_shard && g > 0 ? 0 : apps[(size_t)g].frames_integrated_;
    last_id = std::max(last_id, apps[(size_t)g].frame_id_);
  }
  if (rc == 0) {
    // every worker holds a disjoint set of finished units (--shard unit, or the frame-split merge left distributed): one list in key order
    if (gpus > 1 && (unit_shard || (merge && merge_root == ER_MERGE_DISTRIBUTED))) { if (!SaveWorldSharded(apps, app.pcd_filename_)) rc = 1; }
    else if (!apps[(size_t)(merge && merge_root >= 0 ? merge_root : 0)].SaveWorld()) rc = 1;
  }
  stage_done("SaveWorld");
  std::cout << "Total " << last_id << " frames processed." << std::endl;
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  std::cerr << "Integrate All took " << ms << "ms." << std::endl;
  if (ms > 0 && frames_integrated > 0)
    std::cerr << frames_integrated << " frames integrated, " << 1000.0 * frames_integrated / ms << " frames/s end to end (incl. file I/O)" << std::endl;
  for (er_comm_t c : comms) er_comm_destroy(c);
  // world.pcd is closed: leave without freeing the volumes or running the HIP runtime's teardown (~0.1 s a pipeline script would wait for; the
  // driver reclaims the device memory of a process that exits).
  fflush(nullptr);
  _exit(rc);
}
