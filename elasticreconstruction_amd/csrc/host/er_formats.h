// er_formats.h -- the file formats of the ElasticReconstruction pipeline for the C++ host programs
// (header-only, plain C++17 + zlib).  These files are the reference's real API; every
// reader/writer here is byte-compatible with the reference code it cites:
//   .log     RGBDTrajectory::LoadFromFile/SaveToFile     Integrate/TSDFVolumeUnit.h:22-63, BuildCorrespondence/Helper.h:19-60
//   .info    RGBDInformation::LoadFromFile/SaveToFile     BuildCorrespondence/Helper.h:74-119
//   .ctr     ControlGrid::Load                            Integrate/ControlGrid.cpp:15-34
//   camera   CameraParam::LoadFromFile                    Integrate/TSDFVolumeUnit.h:72-95
//   corres_<i>_<j>.txt / .xyzn                            BuildCorrespondence/CorresApp.cpp:100-108,175-184
//   .pcd     v0.7 ascii / binary / binary_compressed, arbitrary field list (cloud_bin_<i>.pcd,
//            CorresApp.cpp:88-90; layout per Matlab_Toolbox/External/matpcl/loadpcd.m:33-224, lzfd.m:21-76)
//            and the binary x/y/z/intensity writer of SaveWorld (TSDFVolume.cpp:104-132)
//   depth    the reference reads only OpenNI devices / .oni files (Integrate.cpp:46-60), which cannot exist
//            here; additive sources: a raw stream of 640x480 little-endian uint16 frames, or a list of
//            16-bit grayscale PNG files (decoded with zlib; libpng headers are not installed).
#pragma once

#include <zlib.h>

#include <algorithm>

#include <cmath>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace erfmt {

struct FramedTransformation {
  int id1 = 0, id2 = 0, frame = 0;
  double T[16];
};

inline bool file_exists(const std::string& p) {
  if (p.empty()) return false;
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) return false;
  fclose(f);
  return true;
}

// RGBDTrajectory::LoadFromFile: fgets 1024-byte lines; a line starting with '#' is skipped where a header
// is expected; header "%d %d %d", then four rows "%lf %lf %lf %lf".
inline bool load_log(const std::string& path, std::vector<FramedTransformation>& out) {
  out.clear();
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return false;
  char buf[1024];
  while (fgets(buf, 1024, f)) {
    if (strlen(buf) > 0 && buf[0] != '#') {
      FramedTransformation t;
      if (sscanf(buf, "%d %d %d", &t.id1, &t.id2, &t.frame) < 3) continue;
      bool ok = true;
      for (int r = 0; r < 4 && ok; r++) {
        if (!fgets(buf, 1024, f)) { ok = false; break; }
        sscanf(buf, "%lf %lf %lf %lf", &t.T[r * 4], &t.T[r * 4 + 1], &t.T[r * 4 + 2], &t.T[r * 4 + 3]);
      }
      if (!ok) break;
      out.push_back(t);
    }
  }
  fclose(f);
  return true;
}

inline bool save_log(const std::string& path, const std::vector<FramedTransformation>& v) {
  FILE* f = fopen(path.c_str(), "w");
  if (!f) return false;
  for (const auto& t : v) {
    fprintf(f, "%d\t%d\t%d\n", t.id1, t.id2, t.frame);
    for (int r = 0; r < 4; r++) fprintf(f, "%.8f %.8f %.8f %.8f\n", t.T[r * 4], t.T[r * 4 + 1], t.T[r * 4 + 2], t.T[r * 4 + 3]);
  }
  fclose(f);
  return true;
}

struct FramedInformation {
  int id1 = 0, id2 = 0, frame = 0;
  double info[36];
};

inline bool save_info(const std::string& path, const std::vector<FramedInformation>& v) {
  FILE* f = fopen(path.c_str(), "w");
  if (!f) return false;
  for (const auto& t : v) {
    fprintf(f, "%d\t%d\t%d\n", t.id1, t.id2, t.frame);
    for (int r = 0; r < 6; r++)
      fprintf(f, "%.8f %.8f %.8f %.8f %.8f %.8f\n", t.info[r * 6], t.info[r * 6 + 1], t.info[r * 6 + 2], t.info[r * 6 + 3],
              t.info[r * 6 + 4], t.info[r * 6 + 5]);
  }
  fclose(f);
  return true;
}

// CameraParam: fx fy cx cy ICP_trunc integration_trunc, one per line (defaults TSDFVolumeUnit.h:69).
inline void load_camera(const std::string& path, float cam[6]) {
  const float def[6] = {525.0f, 525.0f, 319.5f, 239.5f, 2.5f, 2.5f};
  memcpy(cam, def, sizeof def);
  FILE* f = path.empty() ? nullptr : fopen(path.c_str(), "r");
  if (!f) return;
  char buf[1024];
  int i = 0;
  while (i < 6 && fgets(buf, 1024, f))
    if (strlen(buf) > 0 && buf[0] != '#' && sscanf(buf, "%f", &cam[i]) == 1) i++;
  fclose(f);
  fprintf(stdout, "Camera model set to (fx, fy, cx, cy, icp_trunc, int_trunc):\n\t%.2f, %.2f, %.2f, %.2f, %.2f, %.2f\n", cam[0],
          cam[1], cam[2], cam[3], cam[4], cam[5]);
}

// ControlGrid::Load x num: (res+1)^3 lines "%f %f %f" per grid.
inline bool load_ctr(const std::string& path, int num, int res, std::vector<float>& out) {
  const size_t verts = (size_t)(res + 1) * (res + 1) * (res + 1);
  out.assign((size_t)num * verts * 3, 0.f);
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return false;
  char buf[1024];
  for (size_t i = 0; i < (size_t)num * verts; i++) {
    if (!fgets(buf, 1024, f)) break;
    if (strlen(buf) > 0 && buf[0] != '#') sscanf(buf, "%f %f %f", &out[i * 3], &out[i * 3 + 1], &out[i * 3 + 2]);
  }
  fclose(f);
  return true;
}

// ---------------------------------------------------------------------------------------- PCD v0.7
inline bool lzf_decompress(const uint8_t* in, size_t in_len, std::vector<uint8_t>& out, size_t out_len) {
  out.assign(out_len, 0);
  size_t ip = 0, op = 0;
  while (ip < in_len) {
    unsigned ctrl = in[ip++];
    if (ctrl < 32) {
      size_t ln = ctrl + 1;
      if (op + ln > out_len || ip + ln > in_len) return false;
      memcpy(&out[op], &in[ip], ln);
      ip += ln;
      op += ln;
    } else {
      size_t ln = ctrl >> 5;
      if (ln == 7) { if (ip >= in_len) return false; ln += in[ip++]; }
      if (ip >= in_len) return false;
      size_t back = ((size_t)(ctrl & 0x1F) << 8) + in[ip++] + 1;
      if (back > op || op + ln + 2 > out_len) return false;
      size_t ref = op - back;
      for (size_t k = 0; k < ln + 2; k++) out[op++] = out[ref++];
    }
  }
  return true;
}

struct PcdField { std::string name; int size = 4; char type = 'F'; int count = 1; };

// Loads the named float32 fields of a PCD file (others are ignored; missing ones are filled with NaN).
inline bool load_pcd_fields(const std::string& path, const std::vector<std::string>& want, std::vector<std::vector<float>>& cols,
                            size_t& npts) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  std::vector<uint8_t> raw;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  raw.resize((size_t)sz);
  if (sz > 0 && fread(raw.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return false; }
  fclose(f);
  std::vector<PcdField> fields;
  std::string mode;
  size_t pos = 0, width = 0, height = 1;
  npts = 0;
  bool have_points = false;
  while (pos < raw.size()) {
    size_t end = pos;
    while (end < raw.size() && raw[end] != '\n') end++;
    std::string line((const char*)&raw[pos], end - pos);
    pos = end + 1;
    while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    std::vector<std::string> tok;
    size_t a = 0;
    while (a < line.size()) {
      while (a < line.size() && line[a] == ' ') a++;
      size_t b = a;
      while (b < line.size() && line[b] != ' ') b++;
      if (b > a) tok.push_back(line.substr(a, b - a));
      a = b;
    }
    if (tok.empty()) continue;
    const std::string& k = tok[0];
    if (k == "FIELDS") { fields.resize(tok.size() - 1); for (size_t i = 1; i < tok.size(); i++) fields[i - 1].name = tok[i]; }
    else if (k == "SIZE") { for (size_t i = 1; i < tok.size() && i - 1 < fields.size(); i++) fields[i - 1].size = atoi(tok[i].c_str()); }
    else if (k == "TYPE") { for (size_t i = 1; i < tok.size() && i - 1 < fields.size(); i++) fields[i - 1].type = tok[i][0]; }
    else if (k == "COUNT") { for (size_t i = 1; i < tok.size() && i - 1 < fields.size(); i++) fields[i - 1].count = atoi(tok[i].c_str()); }
    else if (k == "WIDTH" && tok.size() > 1) width = (size_t)atoll(tok[1].c_str());
    else if (k == "HEIGHT" && tok.size() > 1) height = (size_t)atoll(tok[1].c_str());
    else if (k == "POINTS" && tok.size() > 1) { npts = (size_t)atoll(tok[1].c_str()); have_points = true; }
    else if (k == "DATA" && tok.size() > 1) { mode = tok[1]; break; }
  }
  if (!have_points) npts = width * height;
  if (mode.empty() || fields.empty()) return false;
  // untrusted header: sizes / counts must be positive and small, and the point count must fit the file (every format
  // spends at least one byte per point and field), so no product below can overflow or trigger a huge allocation
  size_t rec_check = 0;
  for (const PcdField& fd : fields) {
    if (fd.size <= 0 || fd.size > 8 || fd.count <= 0 || fd.count > 65536) return false;
    rec_check += (size_t)fd.size * (size_t)fd.count;
  }
  if (rec_check == 0 || rec_check > (1u << 24)) return false;
  if (npts > raw.size() * (mode == "binary_compressed" ? 256u : 1u)) return false;      // LZF expands by < 256x
  const float nanv = std::nanf("");
  cols.assign(want.size(), std::vector<float>(npts, nanv));
  std::vector<int> target(fields.size(), -1);
  for (size_t i = 0; i < fields.size(); i++)
    for (size_t w = 0; w < want.size(); w++)
      if (fields[i].name == want[w]) target[i] = (int)w;
  auto conv = [](const uint8_t* p, const PcdField& fd) -> float {
    if (fd.type == 'F' && fd.size == 4) { float v; memcpy(&v, p, 4); return v; }
    if (fd.type == 'F' && fd.size == 8) { double v; memcpy(&v, p, 8); return (float)v; }
    if (fd.type == 'U' && fd.size == 1) return (float)*p;
    if (fd.type == 'U' && fd.size == 2) { uint16_t v; memcpy(&v, p, 2); return (float)v; }
    if (fd.type == 'U' && fd.size == 4) { uint32_t v; memcpy(&v, p, 4); return (float)v; }
    if (fd.type == 'I' && fd.size == 1) return (float)*(const int8_t*)p;
    if (fd.type == 'I' && fd.size == 2) { int16_t v; memcpy(&v, p, 2); return (float)v; }
    if (fd.type == 'I' && fd.size == 4) { int32_t v; memcpy(&v, p, 4); return (float)v; }
    return 0.f;
  };
  if (mode == "ascii") {
    const char* p = (const char*)raw.data() + pos;
    const char* e = (const char*)raw.data() + raw.size();
    std::string txt(p, e);
    char* cur = &txt[0];
    for (size_t i = 0; i < npts; i++)
      for (size_t fi = 0; fi < fields.size(); fi++)
        for (int c = 0; c < fields[fi].count; c++) {
          char* nx = nullptr;
          double v = strtod(cur, &nx);
          if (nx == cur) return false;
          cur = nx;
          if (target[fi] >= 0 && c == 0) cols[(size_t)target[fi]][i] = (float)v;
        }
    return true;
  }
  size_t rec = 0;
  std::vector<size_t> off(fields.size());
  for (size_t fi = 0; fi < fields.size(); fi++) { off[fi] = rec; rec += (size_t)fields[fi].size * fields[fi].count; }
  if (mode == "binary") {
    if (pos + rec * npts > raw.size()) return false;
    for (size_t i = 0; i < npts; i++)
      for (size_t fi = 0; fi < fields.size(); fi++)
        if (target[fi] >= 0) cols[(size_t)target[fi]][i] = conv(&raw[pos + i * rec + off[fi]], fields[fi]);
    return true;
  }
  if (mode == "binary_compressed") {
    if (pos + 8 > raw.size()) return false;
    uint32_t csz, usz;
    memcpy(&csz, &raw[pos], 4);
    memcpy(&usz, &raw[pos + 4], 4);
    if (pos + 8 + csz > raw.size()) return false;
    std::vector<uint8_t> blob;
    if (!lzf_decompress(&raw[pos + 8], csz, blob, usz)) return false;
    size_t base = 0;                                        // field-major (SoA) after decompression
    for (size_t fi = 0; fi < fields.size(); fi++) {
      const size_t stride = (size_t)fields[fi].size * fields[fi].count;
      if (target[fi] >= 0) {
        if (base + stride * npts > blob.size()) return false;
        for (size_t i = 0; i < npts; i++) cols[(size_t)target[fi]][i] = conv(&blob[base + i * stride], fields[fi]);
      }
      base += stride * npts;
    }
    return true;
  }
  return false;
}

// pcl::io::savePCDFile( name, PointCloud<PointXYZI>, binary = true ) layout (TSDFVolume.cpp:130).
inline bool save_pcd_xyzi(const std::string& path, const float* xyzi, size_t n) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  fprintf(f,
          "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\n"
          "COUNT 1 1 1 1\nWIDTH %zu\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA binary\n",
          n, n);
  if (n) fwrite(xyzi, sizeof(float) * 4, n, f);
  fclose(f);
  return true;
}

// LZF stream writer (the format lzf_decompress above reads; liblzf's published container-less format as PCL's
// "DATA binary_compressed" uses it): greedy matcher over a hash of 3-byte strings, back references of 3..264 bytes
// up to 8192 bytes back, literal runs of up to 32 bytes.
inline void lzf_compress(const uint8_t* in, size_t n, std::vector<uint8_t>& out) {
  out.clear();
  out.reserve(n + n / 32 + 8);
  std::vector<int64_t> last((size_t)1 << 16, -1);
  size_t lit = 0;
  auto flush = [&](size_t end) {
    while (lit < end) {
      size_t run = std::min<size_t>(32, end - lit);
      out.push_back((uint8_t)(run - 1));
      out.insert(out.end(), in + lit, in + lit + run);
      lit += run;
    }
  };
  auto slot = [&](size_t i) { return (size_t)((((uint32_t)in[i] << 16 | (uint32_t)in[i + 1] << 8 | in[i + 2]) * 2654435761u) >> 16); };
  size_t i = 0;
  while (i + 2 < n) {
    const size_t h = slot(i);
    const int64_t ref = last[h];
    last[h] = (int64_t)i;
    if (ref >= 0 && i - (size_t)ref <= 8192 && in[ref] == in[i] && in[ref + 1] == in[i + 1] && in[ref + 2] == in[i + 2]) {
      const size_t cap = std::min<size_t>(n - i, 264);
      size_t len = 3;
      while (len < cap && in[(size_t)ref + len] == in[i + len]) len++;
      flush(i);
      const size_t back = i - (size_t)ref - 1, l = len - 2;
      if (l < 7) out.push_back((uint8_t)((l << 5) | (back >> 8)));
      else { out.push_back((uint8_t)((7u << 5) | (back >> 8))); out.push_back((uint8_t)(l - 7)); }
      out.push_back((uint8_t)(back & 0xFF));
      for (size_t k = i + 1; k < i + len && k + 2 < n; k++) last[slot(k)] = (int64_t)k;
      i += len;
      lit = i;
    } else {
      i++;
    }
  }
  flush(n);
}

// pcl::PCDWriter::writeBinaryCompressed( name, PointCloud<PointXYZRGBNormal> ) (OptApp.cpp:921-922): the templated writer's
// packed field list (padding dropped), fields stored one after another (all x, all y, ...), the block LZF-compressed and
// preceded by its compressed and uncompressed sizes.  cols[c] points at n floats of field c.
inline bool save_pcd_compressed(const std::string& path, const std::vector<std::string>& names, const std::vector<const float*>& cols, size_t n) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  std::string fields, sizes, types, counts;
  for (const std::string& nm : names) { fields += " " + nm; sizes += " 4"; types += " F"; counts += " 1"; }
  fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS%s\nSIZE%s\nTYPE%s\nCOUNT%s\nWIDTH %zu\nHEIGHT 1\n"
             "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS %zu\nDATA binary_compressed\n",
          fields.c_str(), sizes.c_str(), types.c_str(), counts.c_str(), n, n);
  std::vector<uint8_t> raw(names.size() * n * 4), packed;
  for (size_t c = 0; c < names.size(); c++)
    if (n) memcpy(&raw[c * n * 4], cols[c], n * 4);
  lzf_compress(raw.data(), raw.size(), packed);
  const uint32_t hdr[2] = {(uint32_t)packed.size(), (uint32_t)raw.size()};
  fwrite(hdr, 4, 2, f);
  if (!packed.empty()) fwrite(packed.data(), 1, packed.size(), f);
  fclose(f);
  return true;
}

// ------------------------------------------------------------------------------------ depth sources
// 16-bit grayscale PNG (what the pipeline's depth PNGs are, Matlab_Toolbox/Core/mrMatchDepthColor.m:15-29):
// non-interlaced, colour type 0, bit depth 16 (or 8); chunks inflated with zlib, scanlines un-filtered here.
inline bool load_png16(const std::string& path, int& w, int& h, std::vector<uint16_t>& px) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  std::vector<uint8_t> raw;
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  raw.resize((size_t)sz);
  if (sz > 0 && fread(raw.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return false; }
  fclose(f);
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (raw.size() < 8 || memcmp(raw.data(), sig, 8) != 0) return false;
  auto be32 = [&](size_t o) { return ((uint32_t)raw[o] << 24) | ((uint32_t)raw[o + 1] << 16) | ((uint32_t)raw[o + 2] << 8) | raw[o + 3]; };
  size_t pos = 8;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat;
  w = h = 0;
  while (pos + 12 <= raw.size()) {
    uint32_t len = be32(pos);
    std::string type((const char*)&raw[pos + 4], 4);
    if (pos + 12 + len > raw.size()) return false;
    const uint8_t* d = &raw[pos + 8];
    if (type == "IHDR") {
      if (len != 13) return false;                                      // untrusted input: a short IHDR would be read past its end
      const uint32_t uw = be32(pos + 8), uh = be32(pos + 12);
      if (uw == 0 || uh == 0 || uw > 16384u || uh > 16384u) return false;   // bound the allocation below (depth sensors: <= 4k x 4k)
      w = (int)uw;
      h = (int)uh;
      depth = d[8]; ctype = d[9]; interlace = d[12];
    } else if (type == "IDAT") {
      idat.insert(idat.end(), d, d + len);
    } else if (type == "IEND") {
      break;
    }
    pos += 12 + len;
  }
  // 8-bit files are refused: the caller treats the values as 16-bit millimetres (an 8-bit image is not a depth map)
  if (w <= 0 || h <= 0 || ctype != 0 || interlace != 0 || depth != 16) return false;
  const size_t bpp = depth / 8, stride = (size_t)w * bpp;
  std::vector<uint8_t> img((stride + 1) * (size_t)h);
  uLongf out_len = (uLongf)img.size();
  if (uncompress(img.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != img.size()) return false;
  px.assign((size_t)w * h, 0);
  std::vector<uint8_t> prev(stride, 0), cur(stride);
  for (int y = 0; y < h; y++) {
    const uint8_t* line = &img[(stride + 1) * (size_t)y];
    const int ft = line[0];
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int v = line[1 + i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) / 2; break;
        case 4: {
          const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: return false;
      }
      cur[i] = (uint8_t)v;
    }
    for (int x = 0; x < w; x++)
      px[(size_t)y * w + x] = depth == 16 ? (uint16_t)((cur[2 * x] << 8) | cur[2 * x + 1]) : (uint16_t)cur[x];
    prev.swap(cur);
  }
  return true;
}

// Minimal 16-bit grayscale PNG writer (zlib stored/deflated, filter 0) used by tests and synthetic data tools.
inline bool save_png16(const std::string& path, int w, int h, const uint16_t* px) {
  std::vector<uint8_t> rawimg(((size_t)w * 2 + 1) * (size_t)h);
  for (int y = 0; y < h; y++) {
    uint8_t* line = &rawimg[((size_t)w * 2 + 1) * (size_t)y];
    line[0] = 0;
    for (int x = 0; x < w; x++) { line[1 + 2 * x] = (uint8_t)(px[(size_t)y * w + x] >> 8); line[2 + 2 * x] = (uint8_t)(px[(size_t)y * w + x] & 0xFF); }
  }
  uLongf clen = compressBound((uLong)rawimg.size());
  std::vector<uint8_t> comp(clen);
  if (compress2(comp.data(), &clen, rawimg.data(), (uLong)rawimg.size(), 3) != Z_OK) return false;
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  fwrite(sig, 1, 8, f);
  auto chunk = [&](const char* type, const uint8_t* d, uint32_t len) {
    uint8_t b[4] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len};
    fwrite(b, 1, 4, f);
    fwrite(type, 1, 4, f);
    if (len) fwrite(d, 1, len, f);
    uLong crc = crc32(0L, (const Bytef*)type, 4);
    if (len) crc = crc32(crc, d, len);
    uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    fwrite(c, 1, 4, f);
  };
  uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w, (uint8_t)(h >> 24), (uint8_t)(h >> 16),
                      (uint8_t)(h >> 8), (uint8_t)h, 16, 0, 0, 0, 0};
  chunk("IHDR", ihdr, 13);
  chunk("IDAT", comp.data(), (uint32_t)clen);
  chunk("IEND", nullptr, 0);
  fclose(f);
  return true;
}

// pcl::console::parse_argument / find_switch equivalents (Integrate.cpp:64-76, BuildCorrespondence.cpp:40-83).
inline int find_argument(int argc, char** argv, const char* name) {
  for (int i = 1; i < argc; i++)
    if (strcmp(argv[i], name) == 0) return i;
  return -1;
}
inline bool find_switch(int argc, char** argv, const char* name) { return find_argument(argc, argv, name) > 0; }
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

// ER_TIMING=1 in the environment: the wall time of each stage of a host program on stderr (bench.py's boundary leg reads them).
inline void stage_done(const char* what) {
  static const bool on = getenv("ER_TIMING") != nullptr;
  static auto last = std::chrono::steady_clock::now();
  const auto now = std::chrono::steady_clock::now();
  if (on) fprintf(stderr, "[timing] %-32s %9.1f ms\n", what, std::chrono::duration<double, std::milli>(now - last).count());
  last = now;
}

}  // namespace erfmt
