// er_corres_stub.h -- TEST INFRASTRUCTURE ONLY (oracle build).
//
// Stand-in for the slice of PCL 1.7 that /root/reference/BuildCorrespondence/{BuildCorrespondence,CorresApp}.cpp and
// /root/reference/GlobalRegistration/RansacCurvature.h touch, so that those sources compile UNMODIFIED, in place, into
// oracle/_ref/ (oracle/Makefile).  Everything the reference wrote itself -- LoadData and the overlap pair generation, the
// Registration pre-check and accept rule, FindCorrespondence with NormalDot and the ratio test, the information matrix,
// Blacklist / Redux / Finalize, RGBDTrajectory / RGBDInformation I/O; getFitness / getInformation / align_redux -- then runs as
// REFERENCE code.  What stays a restatement is exactly the PCL surface below; each piece says what it assumes ([PCL]):
//
//   pcl::io::loadPCDFile<PointT>             own reader: PCD v0.7 ascii / binary / binary_compressed, float32 fields by name
//   pcl::KdTreeFLANN<PointT>                 own EXACT kd-tree; squared distance float32 ((dx*dx)+dy*dy)+dz*dz [PCL: FLANN
//                                            L2_Simple], ties towards the lower index [PCL: unspecified]
//   pcl::transformPointCloudWithNormals      [PCL 1.7 common/impl/transforms.hpp] evaluated in the matrix scalar, stored float
//   pcl::transformPointCloud (Matrix4f)      [PCL] float32, Eigen 4x4 * 4x1 product
//   pcl::IterativeClosestPoint + registration::TransformationEstimationPointToPlaneLLS + DefaultConvergenceCriteria
//                                            [PCL 1.7 registration/impl/icp.hpp, transformation_estimation_point_to_plane_lls.hpp,
//                                            default_convergence_criteria.hpp] restated from memory WITH THE REFERENCE'S OWN
//                                            VENDORED EIGEN (ATA.inverse() * ATb, Matrix4f products) -- an independent second
//                                            statement next to oracle/icp_oracle.cpp (uniform grid + hand-written LU);
//                                            tests/test_corres_reference.py requires the two to agree
//   pcl::Registration<S,T> base, PointNormal, FPFHSignature33, Correspondence(s), PointIndices,
//   registration::CorrespondenceRejector / TransformationEstimationSVD (declarations only)    RansacCurvature.h / PolyRejector.h
//   pcl::console::print_*, PCL_DEBUG, strncat_s, _isnanf                                      MSVC / PCL odds and ends
//
// Nothing here is shipped or linked into the product.
#pragma once

#include "../stub/er_oracle_stub.h"

#include <Eigen/Core>
#include <Eigen/Dense>
#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <limits>
#include <map>
#include <sstream>

#ifndef PCL_DEBUG
#define PCL_DEBUG(...) do { } while (0)
#endif
#ifndef _isnanf
#define _isnanf(x) std::isnan(x)
#endif
#ifndef pcl_isfinite
#define pcl_isfinite(x) std::isfinite(x)
#endif

// CorresApp.cpp:38 (MSVC CRT): append at most `count` characters of src to dst (capacity cap), always terminated.
inline int strncat_s(char* dst, size_t cap, const char* src, size_t count) {
  size_t l = strlen(dst), i = 0;
  for (; i < count && src[i] && l + i + 1 < cap; i++) dst[l + i] = src[i];
  dst[l + i] = 0;
  return 0;
}

namespace pcl {

namespace console {
inline void print_info(const char* fmt, ...) { (void)fmt; }
inline void print_highlight(const char* fmt, ...) { (void)fmt; }
inline void print_error(const char* fmt, ...) { (void)fmt; }
inline void print_warn(const char* fmt, ...) { (void)fmt; }
}  // namespace console

// [PCL] PointNormal: x y z (+pad) | normal (+pad) | curvature (+pad); data[3] = 1.
struct PointNormal {
  union { float data[4]; struct { float x, y, z; }; };
  union { float data_n[4]; float normal[3]; struct { float normal_x, normal_y, normal_z; }; };
  union { struct { float curvature; }; float data_c[4]; };
  PointNormal() {
    x = y = z = 0.f; data[3] = 1.f;
    normal_x = normal_y = normal_z = data_n[3] = 0.f;
    curvature = 0.f; data_c[1] = data_c[2] = data_c[3] = 0.f;
  }
};
struct FPFHSignature33 { float histogram[33]; };
struct Correspondence {
  int index_query, index_match;
  union { float distance; float weight; };
  Correspondence() : index_query(0), index_match(-1), distance(std::numeric_limits<float>::max()) {}
};
typedef std::vector<Correspondence> Correspondences;
typedef boost::shared_ptr<Correspondences> CorrespondencesPtr;
struct PointIndices {
  typedef boost::shared_ptr<PointIndices> Ptr;
  typedef boost::shared_ptr<const PointIndices> ConstPtr;
  std::vector<int> indices;
};
typedef boost::shared_ptr<std::vector<int> > IndicesPtr;

}  // namespace pcl

// ------------------------------------------------------------------------------------------------------------------------
// PCD reader (the reference calls pcl::io::loadPCDFile, CorresApp.cpp:90; format per the reference's own
// Matlab_Toolbox/Core/matpcl/loadpcd.m:33-224 and lzfd.m).  Independent of the product's reader (csrc/host/er_formats.h).
// ------------------------------------------------------------------------------------------------------------------------
namespace er_stub {

inline float* field_ptr(pcl::PointXYZRGBNormal& p, const std::string& f) {
  if (f == "x") return &p.x;
  if (f == "y") return &p.y;
  if (f == "z") return &p.z;
  if (f == "normal_x") return &p.normal_x;
  if (f == "normal_y") return &p.normal_y;
  if (f == "normal_z") return &p.normal_z;
  if (f == "rgb" || f == "rgba") return &p.rgb;
  if (f == "curvature") return &p.curvature;
  return NULL;
}
inline float* field_ptr(pcl::PointNormal& p, const std::string& f) {
  if (f == "x") return &p.x;
  if (f == "y") return &p.y;
  if (f == "z") return &p.z;
  if (f == "normal_x") return &p.normal_x;
  if (f == "normal_y") return &p.normal_y;
  if (f == "normal_z") return &p.normal_z;
  if (f == "curvature") return &p.curvature;
  return NULL;
}

// liblzf decompressor (format: lzfd.m of the reference's Matlab toolbox).
inline bool lzf_decompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len) {
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      ctrl++;
      if (op + ctrl > out_len || ip + ctrl > in_len) return false;
      memcpy(out + op, in + ip, ctrl);
      op += ctrl; ip += ctrl;
    } else {
      unsigned len = ctrl >> 5;
      if (ip >= in_len) return false;
      if (len == 7) len += in[ip++];
      if (ip >= in_len) return false;
      size_t ref = op - ((ctrl & 0x1f) << 8) - 1 - in[ip++];
      len += 2;
      if (ref > op || op + len > out_len) return false;
      for (unsigned i = 0; i < len; i++) out[op + i] = out[ref + i];
      op += len;
    }
  }
  return op == out_len;
}

template <class PointT> int load_pcd(const char* name, pcl::PointCloud<PointT>& cloud) {
  FILE* f = fopen(name, "rb");
  if (!f) return -1;
  std::vector<std::string> fields, types;
  std::vector<int> sizes, counts;
  long npoints = -1, width = 0, height = 1;
  std::string mode;
  char line[4096];
  while (fgets(line, sizeof line, f)) {
    if (line[0] == '#') continue;
    std::istringstream ss(line);
    std::string key, tok;
    ss >> key;
    if (key == "FIELDS" || key == "COLUMNS") while (ss >> tok) fields.push_back(tok);
    else if (key == "SIZE") while (ss >> tok) sizes.push_back(atoi(tok.c_str()));
    else if (key == "TYPE") while (ss >> tok) types.push_back(tok);
    else if (key == "COUNT") while (ss >> tok) counts.push_back(atoi(tok.c_str()));
    else if (key == "WIDTH") ss >> width;
    else if (key == "HEIGHT") ss >> height;
    else if (key == "POINTS") ss >> npoints;
    else if (key == "DATA") { ss >> mode; break; }
  }
  if (npoints < 0) npoints = width * height;
  if (counts.empty()) counts.assign(fields.size(), 1);
  if (mode.empty() || sizes.size() != fields.size() || types.size() != fields.size()) { fclose(f); return -1; }
  std::vector<size_t> off(fields.size());
  size_t stride = 0;
  for (size_t i = 0; i < fields.size(); i++) { off[i] = stride; stride += (size_t)sizes[i] * counts[i]; }
  cloud.points.assign((size_t)npoints, PointT());
  cloud.width = (unsigned)npoints; cloud.height = 1; cloud.is_dense = true;
  if (mode == "ascii") {
    for (long k = 0; k < npoints; k++) {
      if (!fgets(line, sizeof line, f)) { fclose(f); return -1; }
      char* s = line;
      for (size_t i = 0; i < fields.size(); i++)
        for (int c = 0; c < counts[i]; c++) {
          char* e;
          double v = strtod(s, &e);
          if (e == s) {                                   // "nan" variants strtod does not know ([PCL] writes "nan")
            while (*s == ' ' || *s == '\t') s++;
            v = NAN;
            while (*s && *s != ' ' && *s != '\t' && *s != '\n') s++;
          } else s = e;
          float* dst = c == 0 ? field_ptr(cloud.points[k], fields[i]) : NULL;
          if (dst) *dst = (float)v;
        }
    }
  } else {
    std::vector<unsigned char> raw(stride * (size_t)npoints);
    bool soa = false;
    if (mode == "binary") {
      if (fread(raw.data(), 1, raw.size(), f) != raw.size()) { fclose(f); return -1; }
    } else if (mode == "binary_compressed") {
      unsigned csz = 0, usz = 0;
      if (fread(&csz, 4, 1, f) != 1 || fread(&usz, 4, 1, f) != 1 || usz != raw.size()) { fclose(f); return -1; }
      std::vector<unsigned char> comp(csz);
      if (fread(comp.data(), 1, csz, f) != csz || !lzf_decompress(comp.data(), csz, raw.data(), raw.size())) { fclose(f); return -1; }
      soa = true;                                          // fields are stored one after the other
    } else { fclose(f); return -1; }
    size_t soa_off = 0;
    for (size_t i = 0; i < fields.size(); i++) {
      const size_t fsz = (size_t)sizes[i] * counts[i];
      for (long k = 0; k < npoints; k++) {
        float* dst = field_ptr(cloud.points[k], fields[i]);
        if (!dst) continue;
        const unsigned char* src = soa ? &raw[soa_off + fsz * (size_t)k] : &raw[stride * (size_t)k + off[i]];
        if (sizes[i] == 4) memcpy(dst, src, 4);           // F4, or packed rgb as U4: bit copy like PCL's field mapping
        else if (sizes[i] == 8 && types[i] == "F") { double v; memcpy(&v, src, 8); *dst = (float)v; }
      }
      soa_off += fsz * (size_t)npoints;
    }
  }
  fclose(f);
  for (long k = 0; k < npoints; k++)
    if (!std::isfinite(cloud.points[k].x) || !std::isfinite(cloud.points[k].y) || !std::isfinite(cloud.points[k].z)) cloud.is_dense = false;
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------------
// Exact 3-D kd-tree.  Distance of a candidate: float32 ((dx*dx) + dy*dy) + dz*dz with dx = q.x - p.x  [PCL: FLANN L2_Simple
// accumulates diff*diff over the dimensions in order, in the element type].  A far subtree is skipped only if
// fl(q[dim] - split)^2 > best: every point p behind the plane has |fl(q[dim] - p[dim])| >= |fl(q[dim] - split)| (rounding is
// monotone), squares and the sums of non-negative terms are monotone too, so its float distance is >= that bound -- the search
// returns the float-exact minimum, ties broken towards the LOWER original index.
// ------------------------------------------------------------------------------------------------------------------------
class KdTree3f {
 public:
  void build(const std::vector<float>& xyz, const std::vector<int>& ids) {
    p_ = xyz; id_ = ids; nodes_.clear();
    std::vector<int> perm(ids.size());
    for (size_t i = 0; i < perm.size(); i++) perm[i] = (int)i;
    if (!perm.empty()) build_node(perm, 0, (int)perm.size());
    std::vector<float> q(p_.size());
    std::vector<int> qi(id_.size());
    for (size_t i = 0; i < perm.size(); i++) {
      for (int a = 0; a < 3; a++) q[3 * i + a] = p_[3 * (size_t)perm[i] + a];
      qi[i] = id_[perm[i]];
    }
    p_.swap(q); id_.swap(qi);
  }
  bool empty() const { return id_.empty(); }
  // nearest neighbour of q: original index (or -1) and float32 squared distance
  int nearest(const float q[3], float* d2) const {
    int best = -1;
    float bd = std::numeric_limits<float>::infinity();
    if (!nodes_.empty()) search(0, q, best, bd);
    *d2 = bd;
    return best;
  }

 private:
  struct Node { int lo, hi, dim, left, right; float split; };
  std::vector<float> p_;
  std::vector<int> id_;
  std::vector<Node> nodes_;
  int build_node(std::vector<int>& perm, int lo, int hi) {
    const int me = (int)nodes_.size();
    nodes_.push_back(Node());
    Node n; n.lo = lo; n.hi = hi; n.dim = -1; n.left = n.right = -1; n.split = 0.f;
    if (hi - lo > 10) {
      float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
      for (int i = lo; i < hi; i++)
        for (int a = 0; a < 3; a++) {
          const float v = p_[3 * (size_t)perm[i] + a];
          mn[a] = std::min(mn[a], v); mx[a] = std::max(mx[a], v);
        }
      int dim = 0;
      for (int a = 1; a < 3; a++) if (mx[a] - mn[a] > mx[dim] - mn[dim]) dim = a;
      if (mx[dim] > mn[dim]) {
        const int mid = (lo + hi) / 2;
        const std::vector<float>& P = p_;
        std::nth_element(perm.begin() + lo, perm.begin() + mid, perm.begin() + hi,
                         [&P, dim](int a, int b) { return P[3 * (size_t)a + dim] < P[3 * (size_t)b + dim]; });
        n.dim = dim; n.split = p_[3 * (size_t)perm[mid] + dim];
        n.left = build_node(perm, lo, mid);
        n.right = build_node(perm, mid, hi);
      }
    }
    nodes_[me] = n;
    return me;
  }
  void search(int ni, const float q[3], int& best, float& bd) const {
    const Node& n = nodes_[ni];
    if (n.dim < 0) {
      for (int i = n.lo; i < n.hi; i++) {
        const float dx = q[0] - p_[3 * (size_t)i], dy = q[1] - p_[3 * (size_t)i + 1], dz = q[2] - p_[3 * (size_t)i + 2];
        const float d = ((dx * dx) + dy * dy) + dz * dz;
        if (d < bd || (d == bd && id_[i] < best)) { bd = d; best = id_[i]; }
      }
      return;
    }
    const float diff = q[n.dim] - n.split;
    const int first = diff < 0.f ? n.left : n.right, second = diff < 0.f ? n.right : n.left;
    search(first, q, best, bd);
    if (!(diff * diff > bd)) search(second, q, best, bd);
  }
};

}  // namespace er_stub

namespace pcl {

namespace io {
inline int loadPCDFile(const char* name, PointCloud<PointXYZRGBNormal>& c) { return er_stub::load_pcd(name, c); }
inline int loadPCDFile(const std::string& name, PointCloud<PointXYZRGBNormal>& c) { return er_stub::load_pcd(name.c_str(), c); }
inline int loadPCDFile(const std::string& name, PointCloud<PointNormal>& c) { return er_stub::load_pcd(name.c_str(), c); }
}  // namespace io

// [PCL] KdTreeFLANN: points with non-finite coordinates are left out of the index (convertCloudToArray keeps an index map).
template <class PointT> class KdTreeFLANN {
 public:
  typedef boost::shared_ptr<KdTreeFLANN<PointT> > Ptr;
  typedef boost::shared_ptr<const PointCloud<PointT> > PointCloudConstPtr;
  void setInputCloud(const PointCloudConstPtr& cloud) {
    std::vector<float> xyz;
    std::vector<int> ids;
    for (size_t i = 0; i < cloud->points.size(); i++) {
      const PointT& p = cloud->points[i];
      if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
      xyz.push_back(p.x); xyz.push_back(p.y); xyz.push_back(p.z);
      ids.push_back((int)i);
    }
    tree_.build(xyz, ids);
  }
  int nearestKSearch(const PointT& p, int k, std::vector<int>& idx, std::vector<float>& sqd) const {
    if (k != 1 || tree_.empty()) return 0;              // the reference only asks for K = 1 (CorresApp.cpp:130,239; RansacCurvature.h:683)
    const float q[3] = {p.x, p.y, p.z};
    float d;
    const int i = tree_.nearest(q, &d);
    if (i < 0) return 0;
    idx.resize(1); sqd.resize(1);
    idx[0] = i; sqd[0] = d;
    return 1;
  }

 private:
  er_stub::KdTree3f tree_;
};
template <> class KdTreeFLANN<FPFHSignature33> {        // RansacCurvature.h:47,59 (feature matching: out of scope, never called)
 public:
  typedef boost::shared_ptr<KdTreeFLANN<FPFHSignature33> > Ptr;
  void setInputCloud(const boost::shared_ptr<const PointCloud<FPFHSignature33> >&) {}
  int nearestKSearch(const PointCloud<FPFHSignature33>&, int, int, std::vector<int>&, std::vector<float>&) const { return 0; }
};

// [PCL 1.7 common/impl/transforms.hpp] transformPointCloudWithNormals( in, out, Matrix<Scalar,4,4> ): per coordinate
//   static_cast<float>( t(r,0)*x + t(r,1)*y + t(r,2)*z + t(r,3) )   and for the normal without the translation term,
// evaluated in Scalar (double at CorresApp.cpp:145,250), left to right.
template <class PointT, class Scalar>
void transformPointCloudWithNormals(const PointCloud<PointT>& in, PointCloud<PointT>& out, const Eigen::Matrix<Scalar, 4, 4>& t) {
  if (&in != &out) out = in;
  for (size_t i = 0; i < out.points.size(); i++) {
    const Scalar x = in.points[i].x, y = in.points[i].y, z = in.points[i].z;
    const Scalar nx = in.points[i].normal_x, ny = in.points[i].normal_y, nz = in.points[i].normal_z;
    PointT& o = out.points[i];
    o.x = static_cast<float>(t(0, 0) * x + t(0, 1) * y + t(0, 2) * z + t(0, 3));
    o.y = static_cast<float>(t(1, 0) * x + t(1, 1) * y + t(1, 2) * z + t(1, 3));
    o.z = static_cast<float>(t(2, 0) * x + t(2, 1) * y + t(2, 2) * z + t(2, 3));
    o.normal_x = static_cast<float>(t(0, 0) * nx + t(0, 1) * ny + t(0, 2) * nz);
    o.normal_y = static_cast<float>(t(1, 0) * nx + t(1, 1) * ny + t(1, 2) * nz);
    o.normal_z = static_cast<float>(t(2, 0) * nx + t(2, 1) * ny + t(2, 2) * nz);
  }
}
// [PCL 1.7] transformPointCloud( in, out, Matrix4f ): same expression in float, xyz only.
template <class PointT> void transformPointCloud(const PointCloud<PointT>& in, PointCloud<PointT>& out, const Eigen::Matrix4f& t) {
  if (&in != &out) out = in;
  for (size_t i = 0; i < out.points.size(); i++) {
    const float x = in.points[i].x, y = in.points[i].y, z = in.points[i].z;
    PointT& o = out.points[i];
    o.x = static_cast<float>(t(0, 0) * x + t(0, 1) * y + t(0, 2) * z + t(0, 3));
    o.y = static_cast<float>(t(1, 0) * x + t(1, 1) * y + t(1, 2) * z + t(1, 3));
    o.z = static_cast<float>(t(2, 0) * x + t(2, 1) * y + t(2, 2) * z + t(2, 3));
  }
}

namespace registration {

template <class S, class T> class TransformationEstimation {
 public:
  virtual ~TransformationEstimation() {}
  virtual void estimateRigidTransformation(const PointCloud<S>& src, const PointCloud<T>& tgt, const Correspondences& corr,
                                           Eigen::Matrix4f& m) const = 0;
};

// [PCL 1.7 transformation_estimation_point_to_plane_lls.hpp]
template <class S, class T> class TransformationEstimationPointToPlaneLLS : public TransformationEstimation<S, T> {
 public:
  void estimateRigidTransformation(const PointCloud<S>& src, const PointCloud<T>& tgt, const Correspondences& corr,
                                   Eigen::Matrix4f& transformation_matrix) const {
    typedef Eigen::Matrix<double, 6, 1> Vector6d;
    typedef Eigen::Matrix<double, 6, 6> Matrix6d;
    Matrix6d ATA;
    Vector6d ATb;
    ATA.setZero();
    ATb.setZero();
    for (size_t k = 0; k < corr.size(); k++) {
      const S* source_it = &src.points[corr[k].index_query];
      const T* target_it = &tgt.points[corr[k].index_match];
      if (!pcl_isfinite(source_it->x) || !pcl_isfinite(source_it->y) || !pcl_isfinite(source_it->z) ||
          !pcl_isfinite(target_it->x) || !pcl_isfinite(target_it->y) || !pcl_isfinite(target_it->z) ||
          !pcl_isfinite(target_it->normal_x) || !pcl_isfinite(target_it->normal_y) || !pcl_isfinite(target_it->normal_z))
        continue;
      const float& sx = source_it->x;
      const float& sy = source_it->y;
      const float& sz = source_it->z;
      const float& dx = target_it->x;
      const float& dy = target_it->y;
      const float& dz = target_it->z;
      const float& nx = target_it->normal_x;
      const float& ny = target_it->normal_y;
      const float& nz = target_it->normal_z;
      double a = nz * sy - ny * sz;
      double b = nx * sz - nz * sx;
      double c = ny * sx - nx * sy;
      //    0  1  2  3  4  5
      //    6  7  8  9 10 11
      //   12 13 14 15 16 17
      //   18 19 20 21 22 23
      //   24 25 26 27 28 29
      //   30 31 32 33 34 35
      ATA.coeffRef(0) += a * a;
      ATA.coeffRef(1) += a * b;
      ATA.coeffRef(2) += a * c;
      ATA.coeffRef(3) += a * nx;
      ATA.coeffRef(4) += a * ny;
      ATA.coeffRef(5) += a * nz;
      ATA.coeffRef(7) += b * b;
      ATA.coeffRef(8) += b * c;
      ATA.coeffRef(9) += b * nx;
      ATA.coeffRef(10) += b * ny;
      ATA.coeffRef(11) += b * nz;
      ATA.coeffRef(14) += c * c;
      ATA.coeffRef(15) += c * nx;
      ATA.coeffRef(16) += c * ny;
      ATA.coeffRef(17) += c * nz;
      ATA.coeffRef(21) += nx * nx;
      ATA.coeffRef(22) += nx * ny;
      ATA.coeffRef(23) += nx * nz;
      ATA.coeffRef(28) += ny * ny;
      ATA.coeffRef(29) += ny * nz;
      ATA.coeffRef(35) += nz * nz;
      double d = nx * dx + ny * dy + nz * dz - nx * sx - ny * sy - nz * sz;
      ATb.coeffRef(0) += a * d;
      ATb.coeffRef(1) += b * d;
      ATb.coeffRef(2) += c * d;
      ATb.coeffRef(3) += nx * d;
      ATb.coeffRef(4) += ny * d;
      ATb.coeffRef(5) += nz * d;
    }
    ATA.coeffRef(6) = ATA.coeff(1);
    ATA.coeffRef(12) = ATA.coeff(2);
    ATA.coeffRef(13) = ATA.coeff(8);
    ATA.coeffRef(18) = ATA.coeff(3);
    ATA.coeffRef(19) = ATA.coeff(9);
    ATA.coeffRef(20) = ATA.coeff(15);
    ATA.coeffRef(24) = ATA.coeff(4);
    ATA.coeffRef(25) = ATA.coeff(10);
    ATA.coeffRef(26) = ATA.coeff(16);
    ATA.coeffRef(27) = ATA.coeff(22);
    ATA.coeffRef(30) = ATA.coeff(5);
    ATA.coeffRef(31) = ATA.coeff(11);
    ATA.coeffRef(32) = ATA.coeff(17);
    ATA.coeffRef(33) = ATA.coeff(23);
    ATA.coeffRef(34) = ATA.coeff(29);
    // Solve A*x = b
    Vector6d x = static_cast<Vector6d>(ATA.inverse() * ATb);
    // Construct the transformation matrix from x
    const double alpha = x(0), beta = x(1), gamma = x(2);
    transformation_matrix = Eigen::Matrix4f::Zero();
    transformation_matrix(0, 0) = static_cast<float>(cos(gamma) * cos(beta));
    transformation_matrix(0, 1) = static_cast<float>(-sin(gamma) * cos(alpha) + cos(gamma) * sin(beta) * sin(alpha));
    transformation_matrix(0, 2) = static_cast<float>(sin(gamma) * sin(alpha) + cos(gamma) * sin(beta) * cos(alpha));
    transformation_matrix(1, 0) = static_cast<float>(sin(gamma) * cos(beta));
    transformation_matrix(1, 1) = static_cast<float>(cos(gamma) * cos(alpha) + sin(gamma) * sin(beta) * sin(alpha));
    transformation_matrix(1, 2) = static_cast<float>(-cos(gamma) * sin(alpha) + sin(gamma) * sin(beta) * cos(alpha));
    transformation_matrix(2, 0) = static_cast<float>(-sin(beta));
    transformation_matrix(2, 1) = static_cast<float>(cos(beta) * sin(alpha));
    transformation_matrix(2, 2) = static_cast<float>(cos(beta) * cos(alpha));
    transformation_matrix(0, 3) = static_cast<float>(x(3));
    transformation_matrix(1, 3) = static_cast<float>(x(4));
    transformation_matrix(2, 3) = static_cast<float>(x(5));
    transformation_matrix(3, 3) = 1.f;
  }
};

// RansacCurvature.h:67 constructs one; hypothesis generation is out of scope, so it only has to exist.
template <class S, class T> class TransformationEstimationSVD : public TransformationEstimation<S, T> {
 public:
  void estimateRigidTransformation(const PointCloud<S>&, const PointCloud<T>&, const Correspondences&, Eigen::Matrix4f&) const {}
  void estimateRigidTransformation(const PointCloud<S>&, const std::vector<int>&, const PointCloud<T>&, const std::vector<int>&,
                                   Eigen::Matrix4f&) const {}
};

// PolyRejector.h:9-13
class CorrespondenceRejector {
 public:
  virtual ~CorrespondenceRejector() {}
  const std::string& getClassName() const { return rejection_name_; }

 protected:
  std::string rejection_name_;
  boost::shared_ptr<const Correspondences> input_correspondences_;
};

}  // namespace registration

// [PCL 1.7 registration.h] the members RansacCurvature.h:13-24 pulls in, with PCL's defaults.
template <class S, class T> class Registration {
 public:
  typedef PointCloud<S> PointCloudSource;
  typedef typename PointCloudSource::Ptr PointCloudSourcePtr;
  typedef typename PointCloudSource::ConstPtr PointCloudSourceConstPtr;
  typedef PointCloud<T> PointCloudTarget;
  typedef typename PointCloudTarget::ConstPtr PointCloudTargetConstPtr;
  typedef boost::shared_ptr<registration::TransformationEstimation<S, T> > TransformationEstimationPtr;
  Registration()
      : tree_(new KdTreeFLANN<T>), nr_iterations_(0), max_iterations_(10), final_transformation_(Eigen::Matrix4f::Identity()),
        transformation_(Eigen::Matrix4f::Identity()), previous_transformation_(Eigen::Matrix4f::Identity()),
        transformation_epsilon_(0.0), euclidean_fitness_epsilon_(-std::numeric_limits<double>::max()),
        corr_dist_threshold_(std::sqrt(std::numeric_limits<double>::max())), converged_(false), min_number_correspondences_(3) {}
  virtual ~Registration() {}
  void setInputCloud(const PointCloudSourceConstPtr& c) { input_ = c; }      // deprecated alias of setInputSource in 1.7
  void setInputSource(const PointCloudSourceConstPtr& c) { input_ = c; }
  void setInputTarget(const PointCloudTargetConstPtr& c) { target_ = c; tree_->setInputCloud(c); }
  void setMaxCorrespondenceDistance(double d) { corr_dist_threshold_ = d; }
  void setMaximumIterations(int n) { max_iterations_ = n; }
  void setTransformationEpsilon(double e) { transformation_epsilon_ = e; }
  void setEuclideanFitnessEpsilon(double e) { euclidean_fitness_epsilon_ = e; }
  template <class P> void setTransformationEstimation(const boost::shared_ptr<P>& te) { transformation_estimation_ = te; }
  Eigen::Matrix4f getFinalTransformation() const { return final_transformation_; }
  bool hasConverged() const { return converged_; }
  const std::string& getClassName() const { return reg_name_; }
  // [PCL 1.7 registration.hpp] mean squared NN distance of final * input to the target
  double getFitnessScore(double max_range = std::numeric_limits<double>::max()) {
    double fitness_score = 0.0;
    PointCloudSource input_transformed;
    transformPointCloud(*input_, input_transformed, final_transformation_);
    std::vector<int> nn_indices(1);
    std::vector<float> nn_dists(1);
    int nr = 0;
    for (size_t i = 0; i < input_transformed.points.size(); ++i) {
      if (tree_->nearestKSearch(input_transformed.points[i], 1, nn_indices, nn_dists) <= 0) continue;
      if (nn_dists[0] > max_range) continue;
      fitness_score += nn_dists[0];
      nr++;
    }
    return nr > 0 ? fitness_score / nr : std::numeric_limits<double>::max();
  }

 protected:
  std::string reg_name_;
  typename KdTreeFLANN<T>::Ptr tree_;
  int nr_iterations_, max_iterations_;
  PointCloudSourceConstPtr input_;
  PointCloudTargetConstPtr target_;
  Eigen::Matrix4f final_transformation_, transformation_, previous_transformation_;
  double transformation_epsilon_, euclidean_fitness_epsilon_, corr_dist_threshold_;
  bool converged_;
  int min_number_correspondences_;
  TransformationEstimationPtr transformation_estimation_;
};

// [PCL 1.7 registration/impl/icp.hpp + correspondence_estimation.hpp + default_convergence_criteria.hpp], as configured at
// CorresApp.cpp:295-306: no reciprocal correspondences, no rejectors, the user's transformation estimation.
template <class S, class T> class IterativeClosestPoint : public Registration<S, T> {
  typedef Registration<S, T> Base;
  using Base::input_; using Base::target_; using Base::tree_; using Base::nr_iterations_; using Base::max_iterations_;
  using Base::final_transformation_; using Base::transformation_; using Base::previous_transformation_;
  using Base::transformation_epsilon_; using Base::euclidean_fitness_epsilon_; using Base::corr_dist_threshold_;
  using Base::converged_; using Base::min_number_correspondences_; using Base::transformation_estimation_;

 public:
  IterativeClosestPoint() { Base::reg_name_ = "IterativeClosestPoint"; }
  int iterations() const { return nr_iterations_; }     // (stub extra, for the checkers)

  void align(PointCloud<S>& output, const Eigen::Matrix4f& guess = Eigen::Matrix4f::Identity()) {
    typedef PointCloud<S> Cloud;
    boost::shared_ptr<Cloud> input_transformed(new Cloud);
    nr_iterations_ = 0;
    converged_ = false;
    final_transformation_ = guess;
    if (guess != Eigen::Matrix4f::Identity()) {
      input_transformed->resize(input_->size());
      transformCloud(*input_, *input_transformed, guess);
    } else
      *input_transformed = *input_;
    transformation_ = Eigen::Matrix4f::Identity();

    // DefaultConvergenceCriteria state, wired as icp.hpp does
    const double rotation_threshold = 1.0 - transformation_epsilon_;
    const double translation_threshold = transformation_epsilon_;
    const double mse_threshold_relative = euclidean_fitness_epsilon_;
    const double mse_threshold_absolute = 1e-12;
    double correspondences_prev_mse = std::numeric_limits<double>::max();
    Correspondences correspondences;
    const double max_dist_sqr = corr_dist_threshold_ * corr_dist_threshold_;
    std::vector<int> index(1);
    std::vector<float> distance(1);
    do {
      previous_transformation_ = transformation_;
      // CorrespondenceEstimation::determineCorrespondences
      correspondences.clear();
      for (size_t i = 0; i < input_transformed->points.size(); ++i) {
        if (tree_->nearestKSearch(input_transformed->points[i], 1, index, distance) <= 0) continue;
        if (distance[0] > max_dist_sqr) continue;
        Correspondence corr;
        corr.index_query = (int)i;
        corr.index_match = index[0];
        corr.distance = distance[0];
        correspondences.push_back(corr);
      }
      if ((int)correspondences.size() < min_number_correspondences_) {
        PCL_ERROR("[pcl::IterativeClosestPoint::computeTransformation] Not enough correspondences found. Relax your threshold parameters.\n");
        converged_ = false;
        break;
      }
      transformation_estimation_->estimateRigidTransformation(*input_transformed, *target_, correspondences, transformation_);
      transformCloud(*input_transformed, *input_transformed, transformation_);
      final_transformation_ = transformation_ * final_transformation_;
      ++nr_iterations_;
      // DefaultConvergenceCriteria::hasConverged
      converged_ = false;
      if (nr_iterations_ >= max_iterations_) { converged_ = true; break; }
      double cos_angle = 0.5 * (transformation_.coeff(0, 0) + transformation_.coeff(1, 1) + transformation_.coeff(2, 2) - 1);
      double translation_sqr = transformation_.coeff(0, 3) * transformation_.coeff(0, 3) +
                               transformation_.coeff(1, 3) * transformation_.coeff(1, 3) +
                               transformation_.coeff(2, 3) * transformation_.coeff(2, 3);
      if (cos_angle >= rotation_threshold && translation_sqr <= translation_threshold) { converged_ = true; break; }
      double mse = 0;
      for (size_t i = 0; i < correspondences.size(); ++i) mse += correspondences[i].distance;
      mse /= double(correspondences.size());
      if (fabs(mse - correspondences_prev_mse) < mse_threshold_absolute) { converged_ = true; break; }
      if (fabs(mse - correspondences_prev_mse) / correspondences_prev_mse < mse_threshold_relative) { converged_ = true; break; }
      correspondences_prev_mse = mse;
    } while (!converged_);
    transformPointCloud(*input_, output, final_transformation_);
  }

 private:
  // [PCL 1.7 icp.hpp transformCloud] pt_t = tr * pt on (x, y, z, 1) in float32 (Eigen 4x4 * 4x1 product); non-finite points skipped.
  static void transformCloud(const PointCloud<S>& in, PointCloud<S>& out, const Eigen::Matrix4f& tr) {
    if (&in != &out) out = in;
    Eigen::Vector4f pt(0.0f, 0.0f, 0.0f, 1.0f), pt_t;
    for (size_t i = 0; i < in.points.size(); ++i) {
      pt[0] = in.points[i].x; pt[1] = in.points[i].y; pt[2] = in.points[i].z;
      if (!pcl_isfinite(pt[0]) || !pcl_isfinite(pt[1]) || !pcl_isfinite(pt[2])) continue;
      pt_t = tr * pt;
      out.points[i].x = pt_t[0]; out.points[i].y = pt_t[1]; out.points[i].z = pt_t[2];
    }
  }
};

}  // namespace pcl
