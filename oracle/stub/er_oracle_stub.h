// er_oracle_stub.h -- TEST INFRASTRUCTURE ONLY (oracle build).
//
// Minimal single-threaded stand-in for the slice of PCL 1.7 / Boost / OpenNI that the
// reference's Integrate program touches, so that /root/reference/Integrate/*.cpp compile
// UNMODIFIED, in place, into a CPU oracle (oracle/Makefile -> oracle/_ref/).  Nothing here is
// shipped or linked into the product library; only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load what is built from it.
//
// Surface provided (use sites in the reference):
//   PCL_INFO/WARN/ERROR            IntegrateApp.cpp:76,206,211  TSDFVolume.cpp:131  TSDFVolumeUnit.h:92
//   boost::shared_ptr/function/bind/mutex/unique_lock/condition_variable/posix_time/signals2/filesystem
//                                  IntegrateApp.h:47-48  IntegrateApp.cpp:45-67,87-107,140,149,167,173,187
//   openni_wrapper::Image/DepthImage   IntegrateApp.cpp:153-165,177-185
//   pcl::Grabber/ONIGrabber/OpenNIGrabber/PCLException/StopWatch/ScopeTime/console::*/PointXYZI/
//   PointCloud/io::savePCDFile     Integrate.cpp:35-85  IntegrateApp.cpp:27,94,99,105,131  TSDFVolume.cpp:106-130
//
// The fake ONIGrabber reads a raw stream of 640x480 little-endian uint16 depth frames (no header)
// from the path given to -oni; trigger() delivers the next frame synchronously with FrameID 1,2,...
// and does nothing at end of file, so the reference's ten-timeouts exit path fires
// (IntegrateApp.cpp:101-127).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <memory>
#include <functional>
#include <iostream>
#include <chrono>
#include <stdexcept>
#include <sys/stat.h>

#define PCL_INFO(...)  do { if (!::er_stub::quiet()) { fprintf(stdout, __VA_ARGS__); } } while (0)
#define PCL_WARN(...)  do { if (!::er_stub::quiet()) { fprintf(stdout, __VA_ARGS__); } } while (0)
#define PCL_ERROR(...) do { fprintf(stderr, __VA_ARGS__); } while (0)
#ifndef _isnan
#define _isnan(x) std::isnan(x)                       /* MSVC spelling, FragmentOptimizer/PointCloud.cpp:29 */
#endif

namespace er_stub {
inline bool quiet() { static int q = getenv("ER_ORACLE_QUIET") ? 1 : 0; return q != 0; }
}

using namespace std::placeholders;   // boost::bind's _1 _2 _3 live in the global namespace

namespace boost {
template <class T> using shared_ptr = std::shared_ptr<T>;
template <class S> using function = std::function<S>;
using std::bind;

namespace filesystem {
inline bool exists(const std::string& p) { struct stat st; return !p.empty() && ::stat(p.c_str(), &st) == 0; }
}

namespace posix_time { struct millisec { long ms; explicit millisec(long m) : ms(m) {} }; }

// Single-threaded world: locks are no-ops, the "condition variable" is a latched flag.
struct mutex {
  struct scoped_lock { explicit scoped_lock(mutex&) {} };
  struct scoped_try_lock { explicit scoped_try_lock(mutex&) {} bool operator!() const { return false; } };
};
template <class M> struct unique_lock { explicit unique_lock(M&) {} };
struct condition_variable {
  bool flag;
  condition_variable() : flag(false) {}
  void notify_one() { flag = true; }
  template <class L> bool timed_wait(L&, const posix_time::millisec&) { bool f = flag; flag = false; return f; }
};
namespace signals2 { struct connection { void disconnect() {} }; }
}  // namespace boost

namespace openni_wrapper {
struct Image {
  unsigned w, h;
  Image(unsigned w_, unsigned h_) : w(w_), h(h_) {}
  unsigned getWidth() const { return w; }
  unsigned getHeight() const { return h; }
};
struct DepthMetaData { unsigned id; unsigned FrameID() const { return id; } };
struct DepthImage {
  unsigned w, h, id;
  std::vector<unsigned short> px;
  DepthImage(unsigned w_, unsigned h_) : w(w_), h(h_), id(0), px((size_t)w_ * h_) {}
  unsigned getWidth() const { return w; }
  unsigned getHeight() const { return h; }
  void fillDepthImageRaw(unsigned width, unsigned height, unsigned short* out) const {
    if (width != w || height != h) return;
    memcpy(out, px.data(), px.size() * sizeof(unsigned short));
  }
  DepthMetaData getDepthMetaData() const { DepthMetaData m; m.id = id; return m; }
};
}  // namespace openni_wrapper

namespace pcl {

struct PCLException : public std::runtime_error {
  explicit PCLException(const std::string& s) : std::runtime_error(s) {}
};

class Grabber {
 public:
  typedef void(sig_cb_openni_image_depth_image)(const boost::shared_ptr<openni_wrapper::Image>&,
                                                const boost::shared_ptr<openni_wrapper::DepthImage>&, float);
  virtual ~Grabber() {}
  template <class T> bool providesCallback() const { return true; }
  template <class T> boost::signals2::connection registerCallback(const boost::function<T>& cb) {
    cb_ = cb;
    return boost::signals2::connection();
  }
  virtual void start() {}
  virtual void stop() {}

 protected:
  std::function<sig_cb_openni_image_depth_image> cb_;
};

class OpenNIGrabber : public Grabber {
 public:
  explicit OpenNIGrabber(const std::string& = "") { throw PCLException("no OpenNI device in the oracle build"); }
};

class ONIGrabber : public Grabber {
 public:
  ONIGrabber(const std::string& file, bool /*repeat*/, bool /*stream*/) : f_(NULL), next_id_(0) {
    f_ = fopen(file.c_str(), "rb");
    if (!f_) throw PCLException("cannot open raw depth stream " + file);
  }
  ~ONIGrabber() { if (f_) fclose(f_); }
  void trigger() {
    if (!f_ || !cb_) return;
    boost::shared_ptr<openni_wrapper::DepthImage> d(new openni_wrapper::DepthImage(640, 480));
    if (fread(d->px.data(), sizeof(unsigned short), d->px.size(), f_) != d->px.size()) return;  // EOF: stay silent
    d->id = ++next_id_;
    boost::shared_ptr<openni_wrapper::Image> im(new openni_wrapper::Image(640, 480));
    cb_(im, d, 1.0f);
  }

 private:
  FILE* f_;
  unsigned next_id_;
};

class StopWatch {
 public:
  StopWatch() : t0_(std::chrono::steady_clock::now()) {}
  double getTime() const {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count();
  }

 private:
  std::chrono::steady_clock::time_point t0_;
};

class ScopeTime : public StopWatch {
 public:
  explicit ScopeTime(const char* title) : title_(title) {}
  ~ScopeTime() { std::cerr << title_ << " took " << getTime() << "ms.\n"; }

 private:
  std::string title_;
};

namespace console {
// FragmentOptimizer.cpp:44-45
enum VERBOSITY_LEVEL { L_ALWAYS, L_ERROR, L_WARN, L_INFO, L_DEBUG, L_VERBOSE };
inline void setVerbosityLevel(VERBOSITY_LEVEL) {}
inline int find_argument(int argc, char** argv, const char* name) {
  for (int i = 1; i < argc; ++i)
    if (strcmp(argv[i], name) == 0) return i;
  return -1;
}
inline bool find_switch(int argc, char** argv, const char* name) { return find_argument(argc, argv, name) != -1; }
inline int parse_argument(int argc, char** argv, const char* name, std::string& v) {
  int i = find_argument(argc, argv, name);
  if (i > 0 && i + 1 < argc) { v = argv[i + 1]; return i; }
  return -1;
}
inline int parse_argument(int argc, char** argv, const char* name, int& v) {
  int i = find_argument(argc, argv, name);
  if (i > 0 && i + 1 < argc) { v = atoi(argv[i + 1]); return i; }
  return -1;
}
inline int parse_argument(int argc, char** argv, const char* name, double& v) {
  int i = find_argument(argc, argv, name);
  if (i > 0 && i + 1 < argc) { v = atof(argv[i + 1]); return i; }
  return -1;
}
}  // namespace console

struct PointXYZI { float x, y, z, intensity; };
// FragmentOptimizer/PointCloud.cpp:22-40 (LoadFromPCDFile); the oracle feeds points through the XYZN path, so loading is a stub.
struct PointXYZRGBNormal {
  float x, y, z, normal_x, normal_y, normal_z, rgb, curvature;
  PointXYZRGBNormal() : x(0), y(0), z(0), normal_x(0), normal_y(0), normal_z(0), rgb(0), curvature(0) {}   // PCL 1.7 zero-initialises
};

struct PCLHeader { unsigned seq; unsigned long long stamp; std::string frame_id; PCLHeader() : seq(0), stamp(0) {} };
template <class PointT> class PointCloud {
 public:
  typedef boost::shared_ptr<PointCloud<PointT> > Ptr;
  typedef boost::shared_ptr<const PointCloud<PointT> > ConstPtr;
  PCLHeader header;
  std::vector<PointT> points;
  unsigned width, height;
  bool is_dense;
  PointCloud() : width(0), height(0), is_dense(true) {}
  void push_back(const PointT& p) { points.push_back(p); width = (unsigned)points.size(); height = 1; }
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void clear() { points.clear(); width = height = 0; }
  void resize(size_t n) { points.resize(n); width = (unsigned)n; height = 1; }   // RansacCurvature.h:674,761
  PointT& operator[](size_t i) { return points[i]; }
  const PointT& operator[](size_t i) const { return points[i]; }
};

// FragmentOptimizer/OptApp.cpp:921-922 (SavePoints, only with --write_xyzn_sample).  The stand-in stores the same records
// uncompressed ("DATA binary", the packed field list PCL's templated writer emits) -- the checker compares values, not bytes.
struct PCDWriter {
  int writeBinaryCompressed(const std::string& name, const PointCloud<PointXYZRGBNormal>& c) {
    FILE* f = fopen(name.c_str(), "wb");
    if (!f) return -1;
    fprintf(f,
            "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z normal_x normal_y normal_z rgb curvature\n"
            "SIZE 4 4 4 4 4 4 4 4\nTYPE F F F F F F F F\nCOUNT 1 1 1 1 1 1 1 1\nWIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\n"
            "POINTS %zu\nDATA binary\n",
            c.size(), c.size());
    if (c.size()) fwrite(c.points.data(), sizeof(PointXYZRGBNormal), c.size(), f);
    fclose(f);
    return 0;
  }
};

namespace io {
template <class PointT> inline int loadPCDFile(const char*, PointCloud<PointT>&) { return -1; }   // not needed by the checkers
// Binary PCD v0.7 writer for x/y/z/intensity clouds (the only cloud type the Integrate program saves).
inline int savePCDFile(const std::string& name, const PointCloud<PointXYZI>& c, bool /*binary*/) {
  FILE* f = fopen(name.c_str(), "wb");
  if (!f) return -1;
  fprintf(f,
          "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\n"
          "TYPE F F F F\nCOUNT 1 1 1 1\nWIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA binary\n",
          c.size(), c.size());
  if (c.size()) fwrite(c.points.data(), sizeof(PointXYZI), c.size(), f);
  fclose(f);
  return 0;
}
}  // namespace io

}  // namespace pcl
