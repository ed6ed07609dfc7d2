// er_tsdf_math.h -- per-pixel / per-voxel arithmetic of path A, written once as inline functions
// that the HIP kernels in er_tsdf.hip call.  The functions are also compilable for the host (ER_HD
// expands to nothing without hipcc) so tests/hostcheck can exercise the exact same expressions
// against the oracle on a machine without a GPU; the shipped library only ever runs them on device.
//
// Parity rules: every expression keeps the reference's float32/float64 mix
// and evaluation order; the translation unit is built with -ffp-contract=off so no mul+add pair is
// fused; '/' and sqrtf are IEEE correctly rounded (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt); matrix*vector is ((m0*v0 + m1*v1) + m2*v2) + m3*v3.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define ER_HD __host__ __device__ __forceinline__
#else
#define ER_HD inline
#endif

namespace er {

struct Camera {            // CameraParam, TSDFVolumeUnit.h:65-70
  float fx, fy, cx, cy, icp_trunc, integration_trunc;
};

constexpr double kUnitLength = 3.0 / 512.0;   // TSDFVolume.cpp:10
constexpr double kTsdfTrunc = 0.03;           // TSDFVolume.cpp:11
constexpr int kUnitRes = 64;                  // TSDFVolume.cpp:57
constexpr int kUnitVox = 64 * 64 * 64;

// Per-frame constants of IntegrateVolumeUnit (TSDFVolume.cpp:59): rows 0..2 of trans_inv.cast<float>()
// and the translation column of transformation.cast<float>().
struct FrameXform {
  float mi[12];
  float tx, ty, tz;
  float pad;
};

// ---- A3: ScaleDepth, TSDFVolume.cpp:24-32 ------------------------------------------------------
// lambda depends only on the pixel and the camera, so it is tabulated once per volume by the same
// float32 expression; the per-frame part is one multiply and one divide.
ER_HD float scale_lambda(int x, int y, const Camera& c) {
  float xl = ((float)x - c.cx) / c.fx;
  float yl = ((float)y - c.cy) / c.fy;
  return sqrtf((xl * xl + yl * yl) + 1.0f);
}

ER_HD float scale_depth_px(uint16_t d, float lambda, float integration_trunc) {
  float res = ((float)d * lambda) / 1000.f;
  return (res > integration_trunc) ? 0.0f : res;
}

// ---- TSDFVolume::UVD2XYZ, TSDFVolume.h:40-49 (d > 0 is checked by the caller) --------------------
ER_HD void uvd2xyz(int u, int v, uint16_t d, const Camera& c, double& x, double& y, double& z) {
  z = (double)d / 1000.0;
  x = (double)((float)u - c.cx) * z / (double)c.fx;
  y = (double)((float)v - c.cy) * z / (double)c.fy;
}

// ---- A5: unit key of one depth pixel, TSDFVolume.cpp:47-52 ---------------------------------------
// T = rows 0..2 of the float64 pose (12 doubles).  Returns the hash_key (TSDFVolume.h:62-64) or -1
// when a unit index falls outside [0,512) -- coordinates beyond +-96 m, where the reference's key
// would alias another unit; such pixels are skipped and counted by the caller.
ER_HD int touch_key(int u, int v, uint16_t d, const Camera& c, const double* T) {
  double x, y, z;
  uvd2xyz(u, v, d, c, x, y, z);
  double p0 = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
  double p1 = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
  double p2 = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
  double v0 = floor(p0 / kUnitLength + 0.5);
  double v1 = floor(p1 / kUnitLength + 0.5);
  double v2 = floor(p2 / kUnitLength + 0.5);
  const double lo = -(256.0 * 64.0), hi = 256.0 * 64.0;
  if (!(v0 >= lo && v0 < hi && v1 >= lo && v1 < hi && v2 >= lo && v2 < hi)) return -1;
  int xi = ((int)v0 + 256 * 64) / 64;
  int yi = ((int)v1 + 256 * 64) / 64;
  int zi = ((int)v2 + 256 * 64) / 64;
  return xi * 512 * 512 + yi * 512 + zi;
}

// I2F, TSDFVolume.h:66-68: float( (i - 256) * 64 * unit_length_ )
ER_HD float unit_shift(int idx) { return (float)((double)((idx - 256) * 64) * kUnitLength); }

// gridv coordinate, TSDFVolume.cpp:75: float( i * unit_length_ + shift )
ER_HD float grid_coord(int i, float shift) { return (float)((double)i * kUnitLength + (double)shift); }

// ---- A4: one voxel of IntegrateVolumeUnit against one frame, TSDFVolume.cpp:76-94 ----------------
// S/W are the voxel's sdf_/weight_.  Returns true if the voxel was updated.
ER_HD bool voxel_update(float& S, float& W, float g0, float g1, float g2, const FrameXform& f, const Camera& c,
                        int cols, int rows, const float* __restrict__ scaled) {
  float t2 = ((f.mi[8] * g0 + f.mi[9] * g1) + f.mi[10] * g2) + f.mi[11];
  if (!(t2 > 0.0f)) return false;                                        // :77
  float t0 = ((f.mi[0] * g0 + f.mi[1] * g1) + f.mi[2] * g2) + f.mi[3];
  float t1 = ((f.mi[4] * g0 + f.mi[5] * g1) + f.mi[6] * g2) + f.mi[7];
  // :78-79  round( float expr ) with TSDFVolume::round(double) = floor(x + 0.5); the range test is done
  // on the float64 value so out-of-range / NaN never reaches an int conversion.
  double px = floor((double)(t0 * c.fx / t2 + c.cx) + 0.5);
  double py = floor((double)(t1 * c.fy / t2 + c.cy) + 0.5);
  if (!(px >= 0.0 && px < (double)cols && py >= 0.0 && py < (double)rows)) return false;  // :80
  float dp = scaled[(int)py * cols + (int)px];                           // :81
  if (!(dp > 0.001f)) return false;                                      // :82
  float rx = g0 - f.tx, ry = g1 - f.ty, rz = g2 - f.tz;                  // :83-85
  float sdf = dp - sqrtf((rx * rx + ry * ry) + rz * rz);                 // :86
  double sdfd = (double)sdf;
  if (!(sdfd >= -kTsdfTrunc)) return false;                              // :87
  // :88 std::min<float>( 1.0f, sdf / tsdf_trunc_ ).  sdf >= trunc  <=>  the float64 quotient is >= 1,
  // so the (slow) float64 division is only evaluated inside the truncation band; the value is
  // identical either way.
  float tsdf = 1.0f;
  if (sdfd < kTsdfTrunc) {
    float q = (float)(sdfd / kTsdfTrunc);
    tsdf = q < 1.0f ? q : 1.0f;
  }
  S = (S * W + tsdf) / (W + 1.0f);                                       // :93  (w == 1.0f, w * tsdf == tsdf)
  W = W + 1.0f;                                                          // :94
  return true;
}

// ---- A6/A7: one source pixel of Reproject, IntegrateApp.cpp:250-259 ------------------------------
// seg, madj: rows 0..2 of the float64 4x4s (12 doubles each).  ctr: one grid, (res+1)^3 * 3 floats.
// On success returns true and the target cell (row-major pixel index) plus the 16-bit depth dd.
ER_HD bool reproject_px(int u, int v, uint16_t d, const Camera& c, int cols, const double* seg, const double* madj,
                        const float* __restrict__ ctr, int res, float grid_ul, int& cell, uint16_t& dd) {
  double x, y, z;
  uvd2xyz(u, v, d, c, x, y, z);
  double q0 = ((seg[0] * x + seg[1] * y) + seg[2] * z) + seg[3];
  double q1 = ((seg[4] * x + seg[5] * y) + seg[6] * z) + seg[7];
  double q2 = ((seg[8] * x + seg[9] * y) + seg[10] * z) + seg[11];
  // ControlGrid::GetCoordinate, ControlGrid.h:44-81 (float32)
  float a0 = (float)q0 / grid_ul, a1 = (float)q1 / grid_ul, a2 = (float)q2 / grid_ul;
  float f0 = floorf(a0), f1 = floorf(a1), f2 = floorf(a2);
  float fres = (float)res;
  if (!(f0 >= 0.0f && f0 < fres && f1 >= 0.0f && f1 < fres && f2 >= 0.0f && f2 < fres)) return false;
  int c0 = (int)f0, c1 = (int)f1, c2 = (int)f2;
  float r0 = a0 - f0, r1 = a1 - f1, r2 = a2 - f2;
  int n1 = res + 1, n2 = n1 * n1;
  int base = c0 + c1 * n1 + c2 * n2;
  float w0 = 1.0f - r0, w1 = 1.0f - r1, w2 = 1.0f - r2;
  float val[8] = {(w0 * w1) * w2, (w0 * w1) * r2, (w0 * r1) * w2, (w0 * r1) * r2,
                  (r0 * w1) * w2, (r0 * w1) * r2, (r0 * r1) * w2, (r0 * r1) * r2};
  int idx[8] = {base,     base + n2,     base + n1,     base + n1 + n2,
                base + 1, base + 1 + n2, base + 1 + n1, base + 1 + n1 + n2};
  // ControlGrid::GetPosition, ControlGrid.h:82-87: left-to-right float32 sum
  float pos[3];
  for (int a = 0; a < 3; a++) {
    float s = val[0] * ctr[idx[0] * 3 + a];
    for (int t = 1; t < 8; t++) s = s + val[t] * ctr[idx[t] * 3 + a];
    pos[a] = s;
  }
  double pa = (double)pos[0], pb = (double)pos[1], pc = (double)pos[2];
  double e0 = ((madj[0] * pa + madj[1] * pb) + madj[2] * pc) + madj[3];
  double e1 = ((madj[4] * pa + madj[5] * pb) + madj[6] * pc) + madj[7];
  double e2 = ((madj[8] * pa + madj[9] * pb) + madj[10] * pc) + madj[11];
  // TSDFVolume::XYZ2UVD, TSDFVolume.h:51-60 (bounds are the literal 640 x 480)
  if (!(e2 > 0.0)) return false;
  double uu = floor((e0 * (double)c.fx / e2 + (double)c.cx) + 0.5);
  double vv = floor((e1 * (double)c.fy / e2 + (double)c.cy) + 0.5);
  if (!(uu >= 0.0 && uu < 640.0 && vv >= 0.0 && vv < 480.0)) return false;
  double dz = floor(e2 * 1000.0 + 0.5);
  // static_cast<unsigned short>( int ): modular.  Depths whose rounding overflows int32 are
  // undefined behaviour in the reference (> 2147 km); they are dropped here.
  if (!(dz < 2147483648.0)) return false;
  dd = (uint16_t)((uint32_t)(int)dz & 0xFFFFu);
  cell = (int)vv * cols + (int)uu;
  return true;
}

}  // namespace er
