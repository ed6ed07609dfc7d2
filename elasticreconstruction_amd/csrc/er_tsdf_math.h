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
#include <string.h>
#include <cmath>

#if defined(__HIPCC__)
#define ER_HD __host__ __device__ __forceinline__
#else
#define ER_HD inline
#endif

namespace er {

struct Camera {            // CameraParam, TSDFVolumeUnit.h:65-70
  float fx, fy, cx, cy, icp_trunc, integration_trunc;
};

// Host-computed float64 reciprocals used ONLY by the division-free fast paths below (never by an exact path).
struct CameraInv {
  double inv_fx, inv_fy;
  int pp_small;                 // |cx| < 1e6 and |cy| < 1e6 (round_pixel's fast path)
  int pad;
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

// x / 1000.f (TSDFVolume.cpp:30) as q + (x - 1000 q) r, r = RN(1/1000) -- 3 full-rate operations instead of the 11 of the
// IEEE sequence with its half-rate v_rcp_f32 (see band_quotient_core below for the float64 twin).  x = (float)d * lambda with
// an integer d in [0, 65535] and lambda = sqrtf( .. + 1.0f ) >= 1 (or inf / NaN for a degenerate camera), so x is +0, >= 1,
// +inf or NaN; tests/test_hostcheck.py compares core and '/' for EVERY such float: identical bits, given the +inf patch
// (q*1000 - inf is NaN).  Below 1 the core may differ (the residual underflows) -- unreachable, see above.
ER_HD float div1000_core(float x) {
  const float r = 1.0f / 1000.0f;                            // folded, correctly rounded
  const float q = x * r;
  const float e = fmaf(-q, 1000.0f, x);
  const float q2 = fmaf(e, r, q);
  return x == __builtin_inff() ? x : q2;
}

ER_HD float scale_depth_px(uint16_t d, float lambda, float integration_trunc) {
#if defined(__HIP_DEVICE_COMPILE__)
  float res = div1000_core((float)d * lambda);
#else
  float res = ((float)d * lambda) / 1000.f;
#endif
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
ER_HD int key_from_voxels(double v0, double v1, double v2) {
  const double lo = -(256.0 * 64.0), hi = 256.0 * 64.0;
  if (!(v0 >= lo && v0 < hi && v1 >= lo && v1 < hi && v2 >= lo && v2 < hi)) return -1;
  int xi = ((int)v0 + 256 * 64) / 64;
  int yi = ((int)v1 + 256 * 64) / 64;
  int zi = ((int)v2 + 256 * 64) / 64;
  return xi * 512 * 512 + yi * 512 + zi;
}

// The reference's expression, division for division (6 float64 divisions per pixel).
ER_HD int touch_key_exact(int u, int v, uint16_t d, const Camera& c, const double* T) {
  double x, y, z;
  uvd2xyz(u, v, d, c, x, y, z);
  double p0 = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
  double p1 = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
  double p2 = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
  return key_from_voxels(floor(p0 / kUnitLength + 0.5), floor(p1 / kUnitLength + 0.5), floor(p2 / kUnitLength + 0.5));
}

// Division-free evaluation with an exactness guard.  Only the UNIT index of the reference's voxel index is observable:
//   voxel = floor( p/ul + 0.5 ),  unit = (voxel + 256*64) / 64  =  floor( (p/ul + 0.5) / 64 ) + 256
// (floor( floor(w) / 64 ) = floor( w / 64 )), so one fused multiply-add per axis yields W = (p/ul + 0.5)/64 + 256 and the unit
// is floor(W).  p itself is evaluated in ray form, z (T0 (u-cx)/fx + T1 (v-cy)/fy + T2) + T3, with rounded reciprocals: it
// differs from the reference's float64 value by at most ~1e-14 * (|T0 x| + |T1 y| + |T2 z| + |T3|); for magnitudes below 20000
// voxels (117 m) that is < 1e-9 voxel = 1.6e-11 in W.  If W keeps a distance > 1.5e-8 from the nearest integer the reference
// value has the same floor; otherwise (a pixel in 10^7) the exact expression is evaluated.  Beyond 20000 voxels the bound is
// not claimed, but there W lies outside [0, 512) by more than 56 units while the error stays relative (~1e-14): the key is -1
// either way.  NaN and |W| >= 2^52 fail the guard (the fraction is NaN or 0) and take the exact path.
ER_HD int touch_key(int u, int v, uint16_t d, const Camera& c, const CameraInv& ci, const double* T) {
  const double z = (double)d * 0.001;
  const double xr = (double)((float)u - c.cx) * ci.inv_fx;
  const double yr = (double)((float)v - c.cy) * ci.inv_fy;
  const double k = (512.0 / 3.0) / 64.0, k0 = 0.5 / 64.0 + 256.0;
  const double w0 = fma(fma(z, fma(T[0], xr, fma(T[1], yr, T[2])), T[3]), k, k0);
  const double w1 = fma(fma(z, fma(T[4], xr, fma(T[5], yr, T[6])), T[7]), k, k0);
  const double w2 = fma(fma(z, fma(T[8], xr, fma(T[9], yr, T[10])), T[11]), k, k0);
  const double f0 = floor(w0), f1 = floor(w1), f2 = floor(w2);
  const double m = 0.5 - 1.5e-8;
  const bool safe = (fabs((w0 - f0) - 0.5) < m) & (fabs((w1 - f1) - 0.5) < m) & (fabs((w2 - f2) - 0.5) < m);
  if (safe) {
    // a W that passed the guard is no integer, so floor(W) in [0, 512) <=> 0 < W < 512 <=> |W - 256| < 256 (tested on the float64
    // value, before any integer conversion: NaN / overflow safe)
    if (!((fabs(w0 - 256.0) < 256.0) & (fabs(w1 - 256.0) < 256.0) & (fabs(w2 - 256.0) < 256.0))) return -1;
    return ((int)f0 << 18) | ((int)f1 << 9) | (int)f2;
  }
  return touch_key_exact(u, v, d, c, T);
}

// Owner of a volume unit when the volume is sharded BY UNIT over `world` GPUs (bit-exact alternative):
// diagonal stripes of the unit lattice, so the ~30-60 units a frustum touches spread evenly over the GPUs.
ER_HD int unit_owner(int key, int world) {
  const int xi = key >> 18, yi = (key >> 9) & 511, zi = key & 511;
  return (xi + yi + zi) % world;
}

// I2F, TSDFVolume.h:66-68: float( (i - 256) * 64 * unit_length_ )
ER_HD float unit_shift(int idx) { return (float)((double)((idx - 256) * 64) * kUnitLength); }

// gridv coordinate, TSDFVolume.cpp:75: float( i * unit_length_ + shift )
ER_HD float grid_coord(int i, float shift) { return (float)((double)i * kUnitLength + (double)shift); }

// ---- A4: one voxel of IntegrateVolumeUnit against one frame, TSDFVolume.cpp:76-94 ----------------
// Correctly rounded float32 division and square root WITHOUT the range-scaling wrapper.
// hipcc lowers an IEEE '/' to  v_div_scale x2, v_rcp, fma, fma, mul, fma, fma, fma, v_div_fmas, v_div_fixup
// and sqrtf to a scale / v_sqrt / +-1 ulp residual test / unscale / class-fixup sequence.  v_div_scale only
// acts (and v_div_fmas / v_div_fixup are only more than an fma / a move) when the denominator or its reciprocal
// is denormal, |numerator| < ~2^-102, the exponents differ by >= 96, the quotient is denormal, or an operand is
// 0 / inf / NaN; otherwise the bare Newton core below executes the SAME operations and returns the same bits.
// The callers state why their operands stay inside that domain (or why the result does not matter outside);
// tests/hip/arith_check.hip compares cores and operators on the GPU, exhaustively for sqrt.  On the host
// (tests/hostcheck) the plain operators are used.
ER_HD void div2_inrange(float n0, float n1, float d, float& q0, float& q1) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r = __builtin_amdgcn_rcpf(d);
  r = fmaf(fmaf(-d, r, 1.0f), r, r);
  float m = n0 * r;
  m = fmaf(fmaf(-d, m, n0), r, m);
  q0 = fmaf(fmaf(-d, m, n0), r, m);
  m = n1 * r;
  m = fmaf(fmaf(-d, m, n1), r, m);
  q1 = fmaf(fmaf(-d, m, n1), r, m);
#else
  q0 = n0 / d;
  q1 = n1 / d;
#endif
}

ER_HD float div_inrange(float n, float d) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r = __builtin_amdgcn_rcpf(d);
  r = fmaf(fmaf(-d, r, 1.0f), r, r);
  float m = n * r;
  m = fmaf(fmaf(-d, m, n), r, m);
  return fmaf(fmaf(-d, m, n), r, m);
#else
  return n / d;
#endif
}

// Three quotients by one denominator (ControlGrid::GetCoordinate's pt / unit_length_, ControlGrid.h:46-48).
ER_HD void div3_inrange(float n0, float n1, float n2, float d, float& q0, float& q1, float& q2) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r = __builtin_amdgcn_rcpf(d);
  r = fmaf(fmaf(-d, r, 1.0f), r, r);
  float m = n0 * r;
  m = fmaf(fmaf(-d, m, n0), r, m);
  q0 = fmaf(fmaf(-d, m, n0), r, m);
  m = n1 * r;
  m = fmaf(fmaf(-d, m, n1), r, m);
  q1 = fmaf(fmaf(-d, m, n1), r, m);
  m = n2 * r;
  m = fmaf(fmaf(-d, m, n2), r, m);
  q2 = fmaf(fmaf(-d, m, n2), r, m);
#else
  q0 = n0 / d;
  q1 = n1 / d;
  q2 = n2 / d;
#endif
}

ER_HD float sqrt_inrange(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  float s = __builtin_amdgcn_sqrtf(x);                       // <= 1 ulp
  const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
  const float r1 = fmaf(-sm, s, x), r2 = fmaf(-sp, s, x);    // residuals against the two neighbours
  s = (r1 <= 0.0f) ? sm : s;
  s = (r2 > 0.0f) ? sp : s;
  return s;
#else
  return sqrtf(x);
#endif
}

// (double)sdf / tsdf_trunc_ (TSDFVolume.cpp:88) without the IEEE division sequence (v_div_scale x2, v_rcp_f64 -- quarter
// rate --, 5 fma, v_div_fmas, v_div_fixup): with r = RN(1/c), q = RN(x*r), e = x - q*c (exact in one fma) the value
// RN(q + e*r) is the correctly rounded quotient (Markstein's correction step; q is within 1 ulp of x/c).  The theorem's
// side conditions are not argued here but CHECKED: tests/test_hostcheck.py evaluates core and '/' for every float in
// [-0.03f, 0.03f] -- the only arguments voxel_finish passes -- and they agree bit for bit as DOUBLES for all of them except
// x = -0 (core +0, '/' -0), which "dp - dist" with dp > 0.001 cannot produce (x - x is +0 in round-to-nearest).
ER_HD double band_quotient_core(float sdf) {
  const double x = (double)sdf;
  const double r = 1.0 / kTsdfTrunc;                         // folded at compile time, correctly rounded
  const double q = x * r;
  const double e = fma(-q, kTsdfTrunc, x);
  return fma(e, r, q);
}

ER_HD double band_quotient(float sdf) {
#if defined(__HIP_DEVICE_COMPILE__)
  return band_quotient_core(sdf);
#else
  return (double)sdf / kTsdfTrunc;
#endif
}

// TSDFVolume::round (TSDFVolume.h:70-72) of a float expression plus the image-range test of :80, in float32:
//   p = floor( (double)x + 0.5 ),  0 <= p < lim    <=>    -0.5 <= x < lim - 0.5
// (x + 0.5 is exact in float64 and both bounds are floats, so the two float compares decide exactly; NaN fails).
// floor(x + 0.5) itself is NOT computed as a float sum (0.49999997f + 0.5f rounds to 1): for x >= -0.5 it is
// floor(x) + [x - floor(x) >= 0.5], where the float difference is exact for x >= 0 and rounds monotonically
// (to 1.0 at most) for x in [-0.5, 0).  lim_m_half = (float)lim - 0.5f.  p is only meaningful when true is returned.
// Checked against the float64 expression for EVERY float by tests/hip/arith_check.hip.
ER_HD bool pixel_index(float x, float lim_m_half, int& p) {
#if defined(__HIP_DEVICE_COMPILE__)
  // v_cvt_rpi_i32_f32: (int)floor(x + 0.5) with the sum NOT rounded to float32 first -- TSDFVolume::round in one instruction
  // instead of five (floor, subtract, compare, convert, add with carry); 16 of them per (patch, frame) visit of k_integrate.
  // tests/hip/arith_check.hip compares it with the float64 expression for EVERY float (k_pixel_all).
  asm("v_cvt_rpi_i32_f32_e32 %0, %1" : "=v"(p) : "v"(x));
#else
  const float fl = floorf(x);
  p = (int)fl + ((x - fl) >= 0.5f ? 1 : 0);
#endif
  return (x >= -0.5f) & (x < lim_m_half);
}

// S/W are the voxel's sdf_/weight_.  Returns true if the voxel was updated.
//
// Same arithmetic as the reference, arranged for a SIMT machine: the projection is evaluated
// unconditionally (lanes with t2 <= 0 are discarded by the predicate), the five range tests are folded
// into ONE predicate and the depth / truncation tests into a second one, so a voxel costs 2-3 divergent
// regions instead of 6.
//
// Split in two so that a thread can issue the depth gathers of all its rows before it needs the first one
// (k_integrate keeps kRows voxels per thread; the gather's L2 latency then overlaps the other rows' arithmetic):
//   voxel_project  :76-80   the pixel under the voxel, or false
//   voxel_finish   :81-94   everything that depends on the depth sample
ER_HD bool voxel_project(float g0, float g1, float g2, const FrameXform& f, const Camera& c, int cols, int rows, unsigned& pixel) {
  const float t2 = ((f.mi[8] * g0 + f.mi[9] * g1) + f.mi[10] * g2) + f.mi[11];
  const float t0 = ((f.mi[0] * g0 + f.mi[1] * g1) + f.mi[2] * g2) + f.mi[3];
  const float t1 = ((f.mi[4] * g0 + f.mi[5] * g1) + f.mi[6] * g2) + f.mi[7];
  const float n0 = t0 * c.fx, n1 = t1 * c.fy;
  float qu, qv;
  div2_inrange(n0, n1, t2, qu, qv);                                      // shared reciprocal of the depth
  // Depth outside [2^-30, 2^30] m: plain operators.  Inside, the core can only differ from '/' when
  // |n| < 2^-100 or the quotient is denormal (then both quotients are below 2^-72 in magnitude and "+ cx" followed
  // by pixel_index gives the same pixel: the sum is cx itself, or, for cx == 0, a value whose floor and range
  // test do not depend on its low bits; likewise a -0 numerator yields +0 instead of -0) or when |n / t2| >= 2^95 (both land outside the image or are NaN).
  if (!((t2 >= 0x1p-30f) & (t2 <= 0x1p30f))) {
    qu = n0 / t2;
    qv = n1 / t2;
  }
  // :78-80  round( float expr ) and the image-range test, see pixel_index.
  int px, py;
  const bool vx = pixel_index(qu + c.cx, (float)cols - 0.5f, px), vy = pixel_index(qv + c.cy, (float)rows - 0.5f, py);
  pixel = (unsigned)(py * cols + px);
  return (t2 > 0.0f) & vx & vy;                                          // :77,:80
}

// voxel_project for a voxel of a patch that patch_may_update has proven "inside" (below): t2 lies in the division core's
// domain and both image-range tests are true, so only the pixel is computed -- by the very same operations.
ER_HD unsigned voxel_project_inside(float g0, float g1, float g2, const FrameXform& f, const Camera& c, int cols, int rows) {
  const float t2 = ((f.mi[8] * g0 + f.mi[9] * g1) + f.mi[10] * g2) + f.mi[11];
  const float t0 = ((f.mi[0] * g0 + f.mi[1] * g1) + f.mi[2] * g2) + f.mi[3];
  const float t1 = ((f.mi[4] * g0 + f.mi[5] * g1) + f.mi[6] * g2) + f.mi[7];
  float qu, qv;
  div2_inrange(t0 * c.fx, t1 * c.fy, t2, qu, qv);
  int px, py;
  (void)pixel_index(qu + c.cx, (float)cols - 0.5f, px);
  (void)pixel_index(qv + c.cy, (float)rows - 0.5f, py);
  return (unsigned)(py * cols + px);
}

// Squared camera distance of a voxel, TSDFVolume.cpp:83-85 (the argument of the :85 square root).
ER_HD float voxel_dist2(float g0, float g1, float g2, const FrameXform& f) {
  const float rx = g0 - f.tx, ry = g1 - f.ty, rz = g2 - f.tz;            // :83-85
  return (rx * rx + ry * ry) + rz * rz;
}

// voxel_finish with the squared distance d2 = voxel_dist2(..) already evaluated.
ER_HD bool voxel_finish_d2(float& S, float& W, float dp, float d2) {
  // No range guard: for d2 < 2^-96 (voxel within 4e-15 m of the camera centre; hipcc's sqrtf would rescale)
  // the core still returns a non-negative value below 1e-14, and "dp - dist" with dp > 0.001 (the only case
  // that survives the next test) equals dp for any dist below half an ulp of dp (>= 2.9e-11).  +inf and
  // every finite d2 above that go through the core unchanged.
  const float dist = sqrt_inrange(d2);
  const float sdf = dp - dist;                                           // :86
  // :82,:87  "sdf >= -tsdf_trunc_" compares the float with the DOUBLE 0.03.  (float)0.03 = 0.0299999993 lies
  // below 0.03 and is the float nearest to it, so for a float sdf:  sdf >= -0.03 <=> sdf >= -0.03f  and
  // sdf < 0.03 <=> sdf <= 0.03f -- float compares, no conversion on the common path.
  static_assert((double)0.03f < kTsdfTrunc && (double)0.030000003f > kTsdfTrunc, "float neighbours of tsdf_trunc_");
  if (!((dp > 0.001f) & (sdf >= -0.03f))) return false;
  // :88 std::min<float>( 1.0f, sdf / tsdf_trunc_ ).  sdf >= trunc  <=>  the float64 quotient is >= 1
  // <=> min(1, q) == 1, so the (slow) float64 division is only evaluated inside the truncation band;
  // the value is identical either way.
  float tsdf = 1.0f;
  if (sdf <= 0.03f) {
    const float q = (float)band_quotient(sdf);
    tsdf = q < 1.0f ? q : 1.0f;
  }
  // :93  (w == 1.0f, w * tsdf == tsdf).  W + 1 is an integer-valued float in [1, 2^25]; the numerator is 0 or at
  // least ~1e-17 in magnitude (|S| <= 1, |tsdf| is 0 or >= 3e-9 because |sdf| is 0 or >= ulp(0.001)) and at most
  // 2^25: always inside the domain of the core.
  S = div_inrange(S * W + tsdf, W + 1.0f);
  W = W + 1.0f;                                                          // :94
  return true;
}

ER_HD bool voxel_finish(float& S, float& W, float dp, float g0, float g1, float g2, const FrameXform& f) {
  return voxel_finish_d2(S, W, dp, voxel_dist2(g0, g1, g2, f));
}

// ---- "sure" classification of one (voxel, frame) WITHOUT the square root --------------------------------------------------
// Nearly every voxel a frame visits lies far in front of the surface (free space: sdf >= trunc, so tsdf = 1) or far behind
// it (sdf < -trunc: no update); only the +-5 voxels around the surface need the value of sdf.  With c = 0.0301f (trunc plus
// 1e-4) and the float values a = fl(dp - c), b = fl(dp + c):
//   free   :  a > 0  and  d2 < fl(a a)   ==>  dist = RN(sqrt d2) <= (dp - c)(1 + 3 2^-24)  ==>  dp - dist >= c - 1.8e-7 dp
//             (evaluated as d2 < fl(a |a|): the product carries the sign of a, so a <= 0 fails without a second compare)
//   behind :            d2 > fl(b b)   ==>  dist >= (dp + c)(1 - 3 2^-24)                ==>  dp - dist <= -c + 1.8e-7 (dp + c)
// (fl(a a) <= a^2 (1 + 2^-24), sqrt and RN are monotonic, a <= (dp - c)(1 + 2^-24); likewise for b.)  For dp < 64 the
// slack 1.8e-7 * 64.03 = 1.2e-5 is an eighth of the 1e-4 margin, so RN(dp - dist) >= 0.03008 > 0.03f (free: :82 and :87
// pass -- dp > c > 0.001 -- and :88 yields exactly 1) or <= -0.03008 < -0.03f (behind: :87 fails).  dp never exceeds
// integration_trunc (scale_depth_px), so k_integrate enables the shortcut when integration_trunc < 64; a NaN dp or d2 fails
// every comparison and is "unsure".  A lane whose projection failed carries dp = 0: b b = 9.06e-4, so it is "behind" (no
// update -- correct) unless it sits within 3 cm of the camera centre, where it is "unsure" and takes the exact path.
// For a free voxel the update (S W + 1) / (W + 1) is EXACTLY 1 when S == 1 (S W = W and W + 1 are exact for W < 2^24,
// x / x = 1; for W >= 2^24 numerator and denominator are the same rounded sum) or W == 0 (0 S + 1 = 1, 1 / 1); k_integrate
// takes the shortcut for a wave only if every lane of its rows is sure and every free lane is trivial in that sense.
constexpr float kSureBand = 0.0301f;
ER_HD void voxel_classify(float dp, float d2, bool& free_sure, bool& behind_sure) {
  const float a = dp - kSureBand, b = dp + kSureBand;
  free_sure = d2 < a * fabsf(a);
  behind_sure = d2 > b * b;
}
ER_HD bool voxel_free_trivial(float S, float W) { return (S == 1.0f) | (W == 0.0f); }

ER_HD bool voxel_update(float& S, float& W, float g0, float g1, float g2, const FrameXform& f, const Camera& c,
                        int cols, int rows, const float* __restrict__ scaled) {
  unsigned pixel;
  if (!voxel_project(g0, g1, g2, f, c, cols, rows, pixel)) return false;
  return voxel_finish(S, W, scaled[pixel], g0, g1, g2, f);               // :81
}

// ---- exact culling of (voxel patch, frame) pairs -------------------------------------------------
// A wave of k_integrate owns a PATCH of kRows x 64 voxels: an axis-aligned rectangle of the world plane
// x = g0, spanning [g1lo,g1hi] x [g2lo,g2hi].  41 % of the (row, frame) visits of the straightforward kernel
// update nothing (outside the frustum, no usable depth under them, or behind the surface by more than the
// truncation).  This test proves -- conservatively, with margins far above float32 rounding -- that NO voxel
// of the patch can be updated by a frame, so the frame can be dropped from the patch's mask without
// changing a single voxel:
//   * all four corners behind the camera plane                  -> every voxel has t2 <= 0 (TSDFVolume.cpp:77)
//   * the pixel bounding box of the corners misses the image    -> the :80 range test fails everywhere
//     (a planar convex patch in front of the camera projects inside the convex hull of its corners)
//   * M = max of the scaled depth over the 32x32-pixel tiles under the box is <= 0.001 -> :82 fails everywhere
//   * M - (distance from the camera centre to the rectangle) < -trunc - 1e-4          -> :87 fails everywhere
// tile_max: per frame, tiles_x * tiles_y floats written by k_prepare, tstride floats apart (1: one frame's tiles in a row; k_integrate keeps the tiles
// FRAME-fastest, tstride = the batch capacity: its lanes are the frames of the batch and mostly read the same tile).  Returns false only if provably dead.
//
// *inside (second verdict, for the frames that stay): true only if EVERY voxel of the patch provably passes the tests of
// voxel_project -- t2 > 0, t2 inside [2^-30, 2^30], -0.5 <= u < cols - 0.5, -0.5 <= v < rows - 0.5 -- so that k_integrate may
// run voxel_project_inside for the patch.  Proof sketch (u = 2^-24; T_k the exact affine forms, t_k their float values):
//   * |t_k - T_k| <= gamma_4 A_k with A_k = sum of the magnitudes of the four terms over the patch; e_k = 2^-21 A_k (= 8u A_k,
//     evaluated in float) bounds it, at the corners and at every voxel alike.  Hence every voxel's t2 lies in
//     [tau, t2max + 2 e2], tau = t2min - 2 e2 (corner values), and the test demands 2^-20 <= tau, t2max + 2 e2 <= 2^20.
//   * With P = T0 fx / T2 (true u - cx):  |fl(fl(t0 fx) / t2) - P| <= (fx e0 + |P| e2) / tau * (1 + 2.01u) + 2.2u |P|  when
//     e2 / tau <= 2^-12 (demanded), and adding cx costs another u (|P| + |cx|): every computed u, corner or voxel, is within
//     s_u = 1.0625 (fx e0 + ua e2) / tau + 2^-20 (ua + |cx| + 1) of its true value, where ua = 1.001 Q + 1 >= max |P| and Q is
//     the largest computed |u_corner - cx| (P is projective-linear on the rectangle, so its extremes are at corners; the
//     corner errors are absorbed by the 1.001 and the + 1 because s_u <= 1/8 is demanded).
//   * A planar rectangle in front of the camera projects into the convex hull of its corners, so every voxel's computed u
//     lies in [umin - 2 s_u, umax + 2 s_u] of the computed corner values; with s_u <= 1/8 the demand 0.5 <= umin and
//     umax <= cols - 1.5 puts it inside [0.25, cols - 1.25].  Same for v.
// NaNs anywhere make a comparison false and the verdict false.  tests/hostcheck cross-checks every "inside" voxel of the golden
// and fuzz scenes against voxel_project itself.
//
// The patch may also be a BOX [g0lo,g0hi] x [g1lo,g1hi] x [g2lo,g2hi] (eight corners): every argument above only uses that the
// patch is convex with its extreme points among the tested corners -- t2 is affine (its extremes over the box are at corners),
// a convex body in front of the camera projects into the convex hull of its corners, P is projective-linear on it -- and the
// error sums A_k take the largest coordinate magnitudes of the patch.
//
// *full (third verdict, optional; needs tile_lo = per 32 x 32 tile the minimum of the scaled depth over ALL its pixels, written by
// k_prepare next to tile_max): true only if the frame updates EVERY voxel of the patch with tsdf = 1, so that k_integrate needs
// neither the projection nor the depth sample nor any arithmetic of the update:
//   * the patch is "inside" (above): every voxel projects into the image, onto a pixel of the tiles under the corner hull
//     (the computed pixel lies within 0.75 px of the corner hull; the tile range takes 1.5 px);
//   * lo = min of tile_lo over those tiles > 0.001: every pixel a voxel can sample carries a usable depth (":82 dp > 0.001"
//     passes), and dp >= lo;
//   * with D = distance from the camera centre to the FARTHEST corner of the patch (dist <= D for every voxel; the float32
//     evaluations of D here and of dist in the update are within 4e-7 D of the true values):  lo - D > trunc + 1e-4 + 1e-6 D
//     =>  sdf = RN(dp - dist) > trunc for every voxel: ":87" passes and ":88" yields exactly 1.
// The update of such a (voxel, frame) is S = (S W + 1) / (W + 1), W = W + 1 -- exactly (1, W + 1) when S == 1 or W == 0.
// NaNs fail a comparison and the verdict.  tests/hostcheck replays it against the full update for every voxel it covers.
ER_HD bool patch_may_update_box(float g0lo, float g0hi, float g1lo, float g1hi, float g2lo, float g2hi, const FrameXform& f, const Camera& c,
                                int cols, int rows, const float* __restrict__ tile_max, int tiles_x, int tiles_y, bool* inside,
                                const float* __restrict__ tile_lo = nullptr, bool* full = nullptr, int lo_shift = 5, int lo_tiles_x = 0,
                                const float* __restrict__ tile_lo_fine = nullptr, int tstride = 1) {
  *inside = false;
  if (full) *full = false;
  float lo_tile = 0.0f;                                   // min of the scaled depth over every pixel a voxel can sample (0: unknown)
  float umin = 3.0e38f, umax = -3.0e38f, vmin = 3.0e38f, vmax = -3.0e38f, t2min = 3.0e38f, t2max = -3.0e38f;
  const int n0 = g0hi != g0lo ? 2 : 1;
  for (int o = 0; o < n0; o++)
  for (int a = 0; a < 2; a++) {
    const float g0 = o ? g0hi : g0lo;
    const float g1 = a ? g1hi : g1lo;
    for (int b = 0; b < 2; b++) {
      const float g2 = b ? g2hi : g2lo;
      const float t2 = ((f.mi[8] * g0 + f.mi[9] * g1) + f.mi[10] * g2) + f.mi[11];
      const float t0 = ((f.mi[0] * g0 + f.mi[1] * g1) + f.mi[2] * g2) + f.mi[3];
      const float t1 = ((f.mi[4] * g0 + f.mi[5] * g1) + f.mi[6] * g2) + f.mi[7];
#if defined(__HIP_DEVICE_COMPILE__)
      // The corner projections feed conservative tests only (margins of 1.5 px; the "inside" slack budgets 16u where 3.3u + 4u
      // are needed), so the 1-ulp hardware reciprocal replaces the two IEEE divisions: 8 fewer division sequences per
      // (patch, frame).  Validated: the whole -m gpu parity suite is bit-exact with it (profiles/r02a_ab_fast_cull.txt),
      // and tests/test_hostcheck.py stresses both verdicts on the CPU with a reciprocal that is off by +-1 and +-2 ulps.
      const float rt2 = __builtin_amdgcn_rcpf(t2);
      const float u = (t0 * c.fx) * rt2 + c.cx, v = (t1 * c.fy) * rt2 + c.cy;
#elif defined(ER_FAST_CULL_HOSTSIM)
      // tests only (tests/test_hostcheck.py): the same expression with a reciprocal that is off by ER_FAST_CULL_HOSTSIM ulps,
      // the accuracy v_rcp_f32 guarantees, so that the verdicts can be stressed on the CPU
      float rt2 = 1.0f / t2;
      for (int s_ = 0; s_ < (ER_FAST_CULL_HOSTSIM < 0 ? -(ER_FAST_CULL_HOSTSIM) : (ER_FAST_CULL_HOSTSIM)); s_++)
        rt2 = nextafterf(rt2, ER_FAST_CULL_HOSTSIM < 0 ? -3.0e38f : 3.0e38f);
      const float u = (t0 * c.fx) * rt2 + c.cx, v = (t1 * c.fy) * rt2 + c.cy;
#else
      const float u = t0 * c.fx / t2 + c.cx, v = t1 * c.fy / t2 + c.cy;
#endif
      t2min = fminf(t2min, t2);
      t2max = fmaxf(t2max, t2);
      umin = fminf(umin, u); umax = fmaxf(umax, u);
      vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
    }
  }
  if (t2max < -1e-3f) return false;                       // whole patch behind the camera
  float dmax_tile = 3.0e38f;                              // upper bound of the scaled depth any voxel can see
  if (t2min > 0.02f) {                                    // hull argument needs the patch clear of the camera plane
    if (umax < -1.5f || umin > (float)cols + 0.5f || vmax < -1.5f || vmin > (float)rows + 0.5f) return false;
    const int x0 = (int)fmaxf(umin - 1.5f, 0.0f) >> 5, x1 = (int)fminf(umax + 1.5f, (float)(cols - 1)) >> 5;
    const int y0 = (int)fmaxf(vmin - 1.5f, 0.0f) >> 5, y1 = (int)fminf(vmax + 1.5f, (float)(rows - 1)) >> 5;
    if ((x1 - x0 + 1) * (y1 - y0 + 1) <= 48) {
      float m = 0.0f, lo = 3.0e38f;
      for (int ty = y0; ty <= y1; ty++)
        for (int tx = x0; tx <= x1; tx++) {
          m = fmaxf(m, tile_max[(ty * tiles_x + tx) * tstride]);
          if (tile_lo) lo = fminf(lo, tile_lo[(ty * tiles_x + tx) * tstride]);
        }
      dmax_tile = m;
      if (tile_lo) lo_tile = lo;
    }
  }
  if (!(dmax_tile > 0.001f)) return false;                // no pixel with usable depth under the patch
  const float dx = f.tx < g0lo ? g0lo - f.tx : (f.tx > g0hi ? f.tx - g0hi : 0.0f);
  const float dy = f.ty < g1lo ? g1lo - f.ty : (f.ty > g1hi ? f.ty - g1hi : 0.0f);
  const float dz = f.tz < g2lo ? g2lo - f.tz : (f.tz > g2hi ? f.tz - g2hi : 0.0f);
  const float dmin = sqrtf((dx * dx + dy * dy) + dz * dz);
  if (dmax_tile - dmin < -(float)kTsdfTrunc - 1e-4f) return false;   // every voxel is behind the surface by more than trunc
  {
    const float a0 = fmaxf(fabsf(g0lo), fabsf(g0hi)), G1 = fmaxf(fabsf(g1lo), fabsf(g1hi)), G2 = fmaxf(fabsf(g2lo), fabsf(g2hi));
    const float e0 = 0x1p-21f * (((fabsf(f.mi[0]) * a0 + fabsf(f.mi[1]) * G1) + fabsf(f.mi[2]) * G2) + fabsf(f.mi[3]));
    const float e1 = 0x1p-21f * (((fabsf(f.mi[4]) * a0 + fabsf(f.mi[5]) * G1) + fabsf(f.mi[6]) * G2) + fabsf(f.mi[7]));
    const float e2 = 0x1p-21f * (((fabsf(f.mi[8]) * a0 + fabsf(f.mi[9]) * G1) + fabsf(f.mi[10]) * G2) + fabsf(f.mi[11]));
    const float tau = t2min - 2.0f * e2;
    const float ua = 1.001f * fmaxf(fabsf(umin - c.cx), fabsf(umax - c.cx)) + 1.0f;
    const float va = 1.001f * fmaxf(fabsf(vmin - c.cy), fabsf(vmax - c.cy)) + 1.0f;
    const float su = 1.0625f * (fabsf(c.fx) * e0 + ua * e2) / tau + 0x1p-20f * ((ua + fabsf(c.cx)) + 1.0f);
    const float sv = 1.0625f * (fabsf(c.fy) * e1 + va * e2) / tau + 0x1p-20f * ((va + fabsf(c.cy)) + 1.0f);
    *inside = (tau >= 0x1p-20f) & (t2max + 2.0f * e2 <= 0x1p20f) & (e2 * 4096.0f <= tau) & (su <= 0.125f) & (sv <= 0.125f) &
              (umin >= 0.5f) & (umax <= (float)cols - 1.5f) & (vmin >= 0.5f) & (vmax <= (float)rows - 1.5f);
  }
  if (full && *inside && tile_lo) {
    const float ex = fmaxf(fabsf(g0lo - f.tx), fabsf(g0hi - f.tx)), ey = fmaxf(fabsf(g1lo - f.ty), fabsf(g1hi - f.ty)),
                ez = fmaxf(fabsf(g2lo - f.tz), fabsf(g2hi - f.tz));
    const float dfar = sqrtf((ex * ex + ey * ey) + ez * ez);
    const float need = ((float)kTsdfTrunc + 1e-4f) + 1e-6f * dfar;
    // first the 32-pixel tiles that were read for the culling anyway ...
    *full = (lo_tile > 0.001f) & (lo_tile - dfar > need);                     // (NaN / inf: false)
    // ... and only if they fail -- typically because ONE pixel of a tile carries no depth (the warp's scatter leaves holes) -- and the
    // patch lies clearly in front of everything under it (the tile MAXIMUM passes the distance test: a patch near or behind the surface
    // can never be full), the minimum over the finer tiles of 2^lo_shift pixels under the hull (round 4; at most 64 of them)
    if (!*full && tile_lo_fine && dmax_tile - dfar > need) {
      const int a0 = (int)fmaxf(umin - 1.5f, 0.0f) >> lo_shift, a1 = (int)fminf(umax + 1.5f, (float)(cols - 1)) >> lo_shift;
      const int b0 = (int)fmaxf(vmin - 1.5f, 0.0f) >> lo_shift, b1 = (int)fminf(vmax + 1.5f, (float)(rows - 1)) >> lo_shift;
      if ((a1 - a0 + 1) * (b1 - b0 + 1) <= 64) {
        float lo = 3.0e38f;
        for (int ty = b0; ty <= b1; ty++)
          for (int tx = a0; tx <= a1; tx++) lo = fminf(lo, tile_lo_fine[(ty * lo_tiles_x + tx) * tstride]);
        *full = (lo > 0.001f) & (lo - dfar > need);
      }
    }
  }
  return true;
}

// The planar patch (x = g0) k_integrate gives a wave by default.
ER_HD bool patch_may_update(float g0, float g1lo, float g1hi, float g2lo, float g2hi, const FrameXform& f, const Camera& c,
                            int cols, int rows, const float* __restrict__ tile_max, int tiles_x, int tiles_y, bool* inside) {
  return patch_may_update_box(g0, g0, g1lo, g1hi, g2lo, g2hi, f, c, cols, rows, tile_max, tiles_x, tiles_y, inside);
}

// ---- A6/A7: one source pixel of Reproject, IntegrateApp.cpp:250-259 ------------------------------
// float64 reciprocal to ~1 ulp without the IEEE division sequence (fast paths only).
ER_HD double fast_rcp64(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}

// Stage 1: the float32 fragment-cube coordinates Vector3f( seg * UVD2XYZ(u,v,d) ) (IntegrateApp.cpp:250-255).
// Only the float32 ROUNDING of each float64 coordinate is observable.  The division-free value q' differs
// from the reference's float64 value by < delta_r = 4e-15 * (|s_r0| Xmax + |s_r1| Ymax + |s_r2| Zmax + |s_r3|),
// a per-frame bound the host computes from the camera and the 16-bit depth range (seg[12..14]); when
// q' - delta and q' + delta round to the same float, so does the reference value (rounding is monotonic).
// Otherwise (a few pixels per million) the exact expression with its three divisions is evaluated.
ER_HD void cube_coords(int u, int v, uint16_t d, const Camera& c, const CameraInv& ci, const double* seg, float out[3]) {
  // Ray form: seg (x, y, z) = z (s0 (u - cx)/fx + s1 (v - cy)/fy + s2) + s3 -- 2 + 3 x 3 float64 operations.  Error against
  // the reference's float64 value, in ulps of the summed magnitudes |s0| |x| + |s1| |y| + |s2| |z| + |s3|: z carries <= 2
  // (the product with the rounded 0.001), the ray component <= 3.5 (rounded reciprocal focal length, three fused operations),
  // the last fma 0.5, the reference's own sequence <= 5.5: <= 12 of the 18 the bound delta budgets.
  const double z = (double)d * 0.001;
  const double xr = (double)((float)u - c.cx) * ci.inv_fx;
  const double yr = (double)((float)v - c.cy) * ci.inv_fy;
  bool safe = true;
  for (int r = 0; r < 3; r++) {
    const double q = fma(z, fma(seg[4 * r], xr, fma(seg[4 * r + 1], yr, seg[4 * r + 2])), seg[4 * r + 3]);
    const double delta = seg[12 + r];
    // rounding is monotonic: when q - delta and q + delta round to the same float, so do q and the reference value between them
    const float lo = (float)(q - delta), hi = (float)(q + delta);
    safe = safe & (lo == hi);
    out[r] = lo;
  }
  if (safe) return;
  double xe, ye, ze;
  uvd2xyz(u, v, d, c, xe, ye, ze);
  for (int r = 0; r < 3; r++) out[r] = (float)(((seg[4 * r] * xe + seg[4 * r + 1] * ye) + seg[4 * r + 2] * ze) + seg[4 * r + 3]);
}

// Host side of the bound above (cols x rows image, depth <= 65535 mm).
inline void cube_coord_deltas(const double* seg, const Camera& c, int cols, int rows, double out3[3]) {
  const double zmax = 65.535;
  const double xa = fmax(fabs(0.0 - (double)c.cx), fabs((double)(cols - 1) - (double)c.cx)) * zmax / (double)c.fx;
  const double ya = fmax(fabs(0.0 - (double)c.cy), fabs((double)(rows - 1) - (double)c.cy)) * zmax / (double)c.fy;
  for (int r = 0; r < 3; r++)
    out3[r] = 4e-15 * (fabs(seg[4 * r]) * xa + fabs(seg[4 * r + 1]) * ya + fabs(seg[4 * r + 2]) * zmax + fabs(seg[4 * r + 3]));
}

// Stage 3: TSDFVolume::round( x * f / z + c ) of XYZ2UVD (TSDFVolume.h:53-54) as an integer-valued double.
// Fast path: one shared reciprocal of z and one fma; for |w| and |cc| below 1e6 the quotient is below 2.1e6 in magnitude and
// the approximate w = quotient + cc + 0.5 (cc + 0.5 is exact for a float cc of that size) is within 1e-8 of the reference's,
// so if it stays > 1e-6 away from an integer the floor is the same.
// cc_small = |cc| < 1e6, a property of the camera the caller evaluates once.
ER_HD double round_pixel(double e, double f, double e2, double rcp_e2, double cc, bool cc_small) {
  const double w = fma(e * f, rcp_e2, cc + 0.5);
  const double fl = floor(w), fr = w - fl;
  if (cc_small && fabs(w) < 1e6 && fr > 1e-6 && fr < 1.0 - 1e-6) return fl;
  return floor((e * f / e2 + cc) + 0.5);
}

// The lattice as ControlGrid keeps it (3 floats per vertex).  Vertices are addressed by UNSIGNED BYTE offsets from the
// (wave-uniform) lattice pointer: one 32-bit add per vertex and a scalar-base load, instead of a sign extension, a multiplication
// by the vertex size and a 64-bit add each.
ER_HD unsigned lattice_stride(const float*) { return 12u; }
ER_HD void lattice_vertex(const float* __restrict__ ctr, unsigned byte_off, float v[3]) {
  const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(ctr) + byte_off);
  v[0] = p[0]; v[1] = p[1]; v[2] = p[2];
}

// seg: rows 0..2 of the float64 4x4 followed by the three rounding bounds of cube_coords (15 doubles used, stride 16);
// madj: rows 0..2 (12 doubles).  ctr: one grid, (res+1)^3 * 3 floats.
// On success returns true and the target cell (row-major pixel index) plus the 16-bit depth dd.
#ifndef ER_LATTICE_SCALAR
#define ER_LATTICE_SCALAR 1     // reproject_px on the device: the lattice cell's eight vertices through the scalar cache when the wave lies in one cell
#endif
template <typename Lattice>
ER_HD bool reproject_px(int u, int v, uint16_t d, const Camera& c, const CameraInv& ci, int cols, int rows, const double* seg,
                        const double* madj, const Lattice* __restrict__ ctr, int res, float grid_ul, int& cell, uint16_t& dd,
                        double* e_out = nullptr) {
  float pt[3];
  cube_coords(u, v, d, c, ci, seg, pt);
  // ControlGrid::GetCoordinate, ControlGrid.h:44-81 (float32)
  // pt / unit_length_ with ONE reciprocal: for a lattice spacing in [2^-30, 2^30] the core can differ from '/' only when
  // |pt| < 2^-100 (both quotients are then below 2^-69: corner 0 or -1 alike, and a residual that vanishes in 1 - r and
  // in the weighted sums) or |pt / unit_length_| >= 2^95 (both far outside the lattice, or NaN, and rejected below).
  float a0, a1, a2;
  if ((grid_ul >= 0x1p-30f) & (grid_ul <= 0x1p30f)) {
    div3_inrange(pt[0], pt[1], pt[2], grid_ul, a0, a1, a2);
  } else {
    a0 = pt[0] / grid_ul;
    a1 = pt[1] / grid_ul;
    a2 = pt[2] / grid_ul;
  }
  float f0 = floorf(a0), f1 = floorf(a1), f2 = floorf(a2);
  float fres = (float)res;
  if (!(f0 >= 0.0f && f0 < fres && f1 >= 0.0f && f1 < fres && f2 >= 0.0f && f2 < fres)) return false;
  int c0 = (int)f0, c1 = (int)f1, c2 = (int)f2;
  float r0 = a0 - f0, r1 = a1 - f1, r2 = a2 - f2;
  int n1 = res + 1, n2 = n1 * n1;
  const unsigned vs = lattice_stride(ctr);                                 // bytes per vertex
  const unsigned base = (unsigned)(c0 + c1 * n1 + c2 * n2) * vs, s1 = (unsigned)n1 * vs, s2 = (unsigned)n2 * vs;
  float w0 = 1.0f - r0, w1 = 1.0f - r1, w2 = 1.0f - r2;
  float val[8] = {(w0 * w1) * w2, (w0 * w1) * r2, (w0 * r1) * w2, (w0 * r1) * r2,
                  (r0 * w1) * w2, (r0 * w1) * r2, (r0 * r1) * w2, (r0 * r1) * r2};
  unsigned idx[8] = {base,      base + s2,      base + s1,      base + s1 + s2,
                     base + vs, base + vs + s2, base + vs + s1, base + vs + s1 + s2};
  // ControlGrid::GetPosition, ControlGrid.h:82-87: left-to-right float32 sum
  float pos[3], vt[3];
#if defined(__HIP_DEVICE_COMPILE__) && ER_LATTICE_SCALAR
  // The 64 pixels of a wave nearly always lie in ONE lattice cell (a cell is 37.5 cm wide): then the eight vertices are the same for every lane and come
  // through the scalar cache (a wave-uniform offset makes the loads scalar) instead of eight 12-byte gathers from the vector L1.  Same values, same sums.
  {
    const unsigned ub = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    if (__builtin_amdgcn_ballot_w64(base != ub) == 0ull) {
      const unsigned uidx[8] = {ub,      ub + s2,      ub + s1,      ub + s1 + s2,
                                ub + vs, ub + vs + s2, ub + vs + s1, ub + vs + s1 + s2};
      lattice_vertex(ctr, uidx[0], vt);
      pos[0] = val[0] * vt[0];
      pos[1] = val[0] * vt[1];
      pos[2] = val[0] * vt[2];
      for (int t = 1; t < 8; t++) {
        lattice_vertex(ctr, uidx[t], vt);
        pos[0] = pos[0] + val[t] * vt[0];
        pos[1] = pos[1] + val[t] * vt[1];
        pos[2] = pos[2] + val[t] * vt[2];
      }
      goto blended;
    }
  }
#endif
  lattice_vertex(ctr, idx[0], vt);
  pos[0] = val[0] * vt[0];
  pos[1] = val[0] * vt[1];
  pos[2] = val[0] * vt[2];
  for (int t = 1; t < 8; t++) {
    lattice_vertex(ctr, idx[t], vt);
    pos[0] = pos[0] + val[t] * vt[0];
    pos[1] = pos[1] + val[t] * vt[1];
    pos[2] = pos[2] + val[t] * vt[2];
  }
#if defined(__HIP_DEVICE_COMPILE__) && ER_LATTICE_SCALAR
blended:
#endif
  double pa = (double)pos[0], pb = (double)pos[1], pc = (double)pos[2];
  double e0 = ((madj[0] * pa + madj[1] * pb) + madj[2] * pc) + madj[3];
  double e1 = ((madj[4] * pa + madj[5] * pb) + madj[6] * pc) + madj[7];
  double e2 = ((madj[8] * pa + madj[9] * pb) + madj[10] * pc) + madj[11];
  // TSDFVolume::XYZ2UVD, TSDFVolume.h:51-60.  The reference's bounds are the literal 640 x 480 whatever the
  // image size; for images smaller than that its write at vv * cols_ + uu would run past the buffer
  // (undefined behaviour), so the bounds are min(640, cols) x min(480, rows) here -- identical for the
  // 640 x 480 streams the reference supports, and for larger images the 640 x 480 clip is preserved.
  if (e_out) { e_out[0] = e0; e_out[1] = e1; e_out[2] = e2; }              // (tests: the reference's float64 values)
  if (!(e2 > 0.0)) return false;
  const double re = fast_rcp64(e2);
  double uu = round_pixel(e0, (double)c.fx, e2, re, (double)c.cx, ci.pp_small);
  double vv = round_pixel(e1, (double)c.fy, e2, re, (double)c.cy, ci.pp_small);
  const double ulim = cols < 640 ? (double)cols : 640.0, vlim = rows < 480 ? (double)rows : 480.0;
  if (!(uu >= 0.0 && uu < ulim && vv >= 0.0 && vv < vlim)) return false;
  double dz = floor(e2 * 1000.0 + 0.5);
  // static_cast<unsigned short>( int ): modular.  Depths whose rounding overflows int32 are
  // undefined behaviour in the reference (> 2147 km); they are dropped here.
  if (!(dz < 2147483648.0)) return false;
  dd = (uint16_t)((uint32_t)(int)dz & 0xFFFFu);
  cell = (int)vv * cols + (int)uu;
  return true;
}


}  // namespace er
