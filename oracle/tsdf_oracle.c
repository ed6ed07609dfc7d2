/* tsdf_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of path A of the reference
 * (qianyizh/ElasticReconstruction, Integrate/), used by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg as the CHECKER.  The product (elasticreconstruction_amd/csrc) never
 * includes, links or calls anything in oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py checks this file against the
 * reference's own code (oracle/_ref/libref_tsdf.so = /root/reference/Integrate/*.cpp compiled
 * unmodified) byte for byte -- scaled images, re-projected depth images, unit key sets, and every
 * sdf_/weight_ array -- and tests/golden/ holds digests generated from that reference build, so
 * the check also runs where /root/reference is absent.
 *
 * Every function cites the reference lines it follows.  Evaluation order matters: float32 vs
 * float64 per operand is as written in the reference, matrix*vector products use Eigen 3.1.2's
 * coefficient-based order  row.v = ((m0*v0 + m1*v1) + m2*v2) + m3*v3  (no FMA: build with
 * -ffp-contract=off, FragmentOptimizer/external/Eigen/src/Core/products/CoeffBasedProduct.h),
 * and integer conversions use C truncation like the reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define UNIT_RES 64
#define UNIT_VOX (UNIT_RES * UNIT_RES * UNIT_RES)

typedef struct {
  int key, xi, yi, zi;
  float* sdf;    /* TSDFVolumeUnit.h:108 */
  float* weight; /* TSDFVolumeUnit.h:109 */
} oracle_unit;

typedef struct {
  int cols, rows;
  float fx, fy, cx, cy, icp_trunc, integration_trunc; /* CameraParam, TSDFVolumeUnit.h:65-70 */
  double unit_length;                                 /* TSDFVolume.cpp:10 */
  double tsdf_trunc;                                  /* TSDFVolume.cpp:11 */
  oracle_unit* units;                                 /* TSDFVolume.h:27 data_ */
  int n_units, cap_units;
  int* touched;                                       /* per-frame touched_unit, TSDFVolume.cpp:41 */
  int n_touched, cap_touched;
} oracle_volume;

/* ---- small double 4x4 helpers (row-major), Eigen coefficient order ---- */
static void mat4_mul(const double* A, const double* B, double* C) {
  double t[16];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++)
      t[r * 4 + c] = ((A[r * 4 + 0] * B[0 * 4 + c] + A[r * 4 + 1] * B[1 * 4 + c]) + A[r * 4 + 2] * B[2 * 4 + c]) +
                     A[r * 4 + 3] * B[3 * 4 + c];
  memcpy(C, t, sizeof t);
}

/* General 4x4 inverse by cofactors in float64.  The reference calls Eigen's SSE2 4x4 double
 * inverse (Eigen/src/LU/arch/Inverse_SSE.h:163); the two agree to ~1 ulp(double), which only
 * matters where a later float32 cast / pixel rounding sits exactly on a boundary. */
void oracle_mat4_inverse(const double* m, double* out) {
  double inv[16];
  inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
  det = 1.0 / det;
  for (int i = 0; i < 16; i++) out[i] = inv[i] * det;
}

/* TSDFVolume::round, TSDFVolume.h:36-38 */
static int round_half_up(double x) { return (int)floor(x + 0.5); }

/* ---- volume lifetime: TSDFVolume::TSDFVolume (TSDFVolume.cpp:7-13), CameraParam (TSDFVolumeUnit.h:69) ---- */
oracle_volume* oracle_volume_create(int cols, int rows, const float cam6[6]) {
  oracle_volume* v = (oracle_volume*)calloc(1, sizeof(oracle_volume));
  v->cols = cols;
  v->rows = rows;
  v->fx = cam6 ? cam6[0] : 525.0f;
  v->fy = cam6 ? cam6[1] : 525.0f;
  v->cx = cam6 ? cam6[2] : 319.5f;
  v->cy = cam6 ? cam6[3] : 239.5f;
  v->icp_trunc = cam6 ? cam6[4] : 2.5f;
  v->integration_trunc = cam6 ? cam6[5] : 2.5f;
  v->unit_length = 3.0 / 512.0;
  v->tsdf_trunc = 0.03;
  return v;
}

void oracle_volume_destroy(oracle_volume* v) {
  if (!v) return;
  for (int i = 0; i < v->n_units; i++) {
    free(v->units[i].sdf);
    free(v->units[i].weight);
  }
  free(v->units);
  free(v->touched);
  free(v);
}

static oracle_unit* find_unit(oracle_volume* v, int key) {
  for (int i = 0; i < v->n_units; i++)
    if (v->units[i].key == key) return &v->units[i];
  return NULL;
}

/* TSDFVolumeUnit::TSDFVolumeUnit, TSDFVolumeUnit.cpp:4-15: two zeroed float[64^3] */
static oracle_unit* add_unit(oracle_volume* v, int key, int xi, int yi, int zi) {
  if (v->n_units == v->cap_units) {
    v->cap_units = v->cap_units ? v->cap_units * 2 : 64;
    v->units = (oracle_unit*)realloc(v->units, sizeof(oracle_unit) * v->cap_units);
  }
  oracle_unit* u = &v->units[v->n_units++];
  u->key = key; u->xi = xi; u->yi = yi; u->zi = zi;
  u->sdf = (float*)calloc(UNIT_VOX, sizeof(float));
  u->weight = (float*)calloc(UNIT_VOX, sizeof(float));
  return u;
}

/* ---- A3: TSDFVolume::ScaleDepth, TSDFVolume.cpp:19-36 (all float32) ---- */
void oracle_scale_depth(const oracle_volume* v, const uint16_t* depth, float* scaled) {
#pragma omp parallel for
  for (int y = 0; y < v->rows; y++) {
    for (int x = 0; x < v->cols; x++) {
      uint16_t d = depth[y * v->cols + x];
      float xl = ((float)x - v->cx) / v->fx;              /* :24  int - float -> float */
      float yl = ((float)y - v->cy) / v->fy;              /* :25 */
      float lambda = sqrtf((xl * xl + yl * yl) + 1.0f);   /* :26  int 1 -> float */
      float res = ((float)d * lambda) / 1000.f;           /* :27  ushort -> int -> float */
      scaled[y * v->cols + x] = (res > v->integration_trunc) ? 0.0f : res; /* :28-32 */
    }
  }
}

/* TSDFVolume::UVD2XYZ, TSDFVolume.h:40-49.  (u - cx_) is int - float = float32; the product with
 * z and the division by fx_ are float64. */
static int uvd2xyz(const oracle_volume* v, int u, int vv, uint16_t d, double* x, double* y, double* z) {
  if (d > 0) {
    *z = d / 1000.0;
    *x = (double)((float)u - v->cx) * (*z) / (double)v->fx;
    *y = (double)((float)vv - v->cy) * (*z) / (double)v->fy;
    return 1;
  }
  return 0;
}

/* TSDFVolume::XYZ2UVD, TSDFVolume.h:51-60.  Bounds are the literal 640x480 of the reference. */
static int xyz2uvd(const oracle_volume* v, double x, double y, double z, int* u, int* vv, uint16_t* d) {
  if (z > 0) {
    *u = round_half_up(x * (double)v->fx / z + (double)v->cx);
    *vv = round_half_up(y * (double)v->fy / z + (double)v->cy);
    *d = (uint16_t)round_half_up(z * 1000.0);
    return (*u >= 0 && *u < 640 && *vv >= 0 && *vv < 480);
  }
  return 0;
}

/* ---- A4: TSDFVolume::IntegrateVolumeUnit, TSDFVolume.cpp:69-102 ---- */
static void integrate_unit(const oracle_volume* v, const float* scaled, const float* M /*trans, float 4x4*/,
                           const float* Mi /*trans_inv*/, oracle_unit* unit, float xs, float ys, float zs) {
  const double ul = v->unit_length;
#pragma omp parallel for
  for (int i = 0; i < UNIT_RES; i++) {
    for (int j = 0; j < UNIT_RES; j++) {
      for (int k = 0; k < UNIT_RES; k++) {
        /* :75 Vector4f gridv( i * unit_length_ + x_shift, ... , 1 ): int*double + float -> double -> float */
        float g0 = (float)((double)i * ul + (double)xs);
        float g1 = (float)((double)j * ul + (double)ys);
        float g2 = (float)((double)k * ul + (double)zs);
        /* :76 tgv = trans_inv * gridv (float32, coefficient order; gridv(3) == 1.f) */
        float t0 = ((Mi[0] * g0 + Mi[1] * g1) + Mi[2] * g2) + Mi[3] * 1.0f;
        float t1 = ((Mi[4] * g0 + Mi[5] * g1) + Mi[6] * g2) + Mi[7] * 1.0f;
        float t2 = ((Mi[8] * g0 + Mi[9] * g1) + Mi[10] * g2) + Mi[11] * 1.0f;
        if (t2 > 0) {                                                     /* :77 */
          int coox = round_half_up((double)(t0 * v->fx / t2 + v->cx));    /* :78 float expr, double round */
          int cooy = round_half_up((double)(t1 * v->fy / t2 + v->cy));    /* :79 */
          if (coox >= 0 && coox < v->cols && cooy >= 0 && cooy < v->rows) { /* :80 */
            float dp = scaled[cooy * v->cols + coox];                     /* :81 */
            if (dp > 0.001f) {                                            /* :82 */
              float rx = g0 - M[3];                                       /* :83-85 */
              float ry = g1 - M[7];
              float rz = g2 - M[11];
              float sdf = dp - sqrtf((rx * rx + ry * ry) + rz * rz);      /* :86 */
              if ((double)sdf >= -v->tsdf_trunc) {                        /* :87 float vs double compare */
                float q = (float)((double)sdf / v->tsdf_trunc);           /* :88 double division -> float arg */
                float tsdf = q < 1.0f ? q : 1.0f;                         /*     std::min<float>( 1.0f, q ) */
                float w = 1.0f;                                           /* :90 */
                int l = (i * UNIT_RES + j) * UNIT_RES + k;                /* :92 */
                unit->sdf[l] = (unit->sdf[l] * unit->weight[l] + w * tsdf) / (unit->weight[l] + w); /* :93 */
                unit->weight[l] += w;                                     /* :94 */
              }
            }
          }
        }
      }
    }
  }
}

/* ---- A5: TSDFVolume::Integrate, TSDFVolume.cpp:38-67.  T is row-major float64 4x4. ---- */
void oracle_integrate(oracle_volume* v, const uint16_t* depth, const float* scaled, const double* T) {
  double Tinv[16];
  oracle_mat4_inverse(T, Tinv);                                           /* :40 */
  float M[16], Mi[16];
  for (int i = 0; i < 16; i++) { M[i] = (float)T[i]; Mi[i] = (float)Tinv[i]; } /* :59 .cast<float>() */
  const double ul = v->unit_length;
  v->n_touched = 0;
  for (int vv = 0; vv < v->rows; vv++) {
    for (int u = 0; u < v->cols; u++) {
      uint16_t d = depth[vv * v->cols + u];
      double x, y, z;
      if (uvd2xyz(v, u, vv, d, &x, &y, &z)) {                             /* :47 */
        double p0 = ((T[0] * x + T[1] * y) + T[2] * z) + T[3] * 1.0;      /* :48 */
        double p1 = ((T[4] * x + T[5] * y) + T[6] * z) + T[7] * 1.0;
        double p2 = ((T[8] * x + T[9] * y) + T[10] * z) + T[11] * 1.0;
        int xi = ((int)floor(p0 / ul + 0.5) + (256 * 64)) / 64;           /* :49-51 C int division */
        int yi = ((int)floor(p1 / ul + 0.5) + (256 * 64)) / 64;
        int zi = ((int)floor(p2 / ul + 0.5) + (256 * 64)) / 64;
        int key = xi * 512 * 512 + yi * 512 + zi;                         /* :52, TSDFVolume.h:62-64 */
        int seen = 0;
        for (int t = v->n_touched - 1; t >= 0; t--)                       /* :53 touched_unit.find */
          if (v->touched[t] == key) { seen = 1; break; }
        if (!seen) {
          if (v->n_touched == v->cap_touched) {
            v->cap_touched = v->cap_touched ? v->cap_touched * 2 : 256;
            v->touched = (int*)realloc(v->touched, sizeof(int) * v->cap_touched);
          }
          v->touched[v->n_touched++] = key;                               /* :54 */
          oracle_unit* unit = find_unit(v, key);
          if (!unit) unit = add_unit(v, key, xi, yi, zi);                 /* :55-57 */
          /* :59 I2F( xi ) = float( (xi - 256) * 64 * unit_length_ ), TSDFVolume.h:66-68 */
          float xs = (float)((double)((xi - 256) * 64) * ul);
          float ys = (float)((double)((yi - 256) * 64) * ul);
          float zs = (float)((double)((zi - 256) * 64) * ul);
          integrate_unit(v, scaled, M, Mi, unit, xs, ys, zs);
        }
      }
    }
  }
}

/* ---- A6: ControlGrid::GetCoordinate / GetPosition, ControlGrid.h:44-87 (float32) ----
 * ctr: (res+1)^3 vertices x 3 floats, vertex index i + j*(res+1) + k*(res+1)^2 (ControlGrid.h:41-43);
 * grid_ul = float(length) / res (ControlGrid.cpp:17-19: length_ is float, unit_length_ = length_ / resolution_). */
static int grid_warp(const float* ctr, int res, float grid_ul, float p0, float p1, float p2, float* pos) {
  float q0 = p0 / grid_ul, q1 = p1 / grid_ul, q2 = p2 / grid_ul;
  int c0 = (int)floorf(q0), c1 = (int)floorf(q1), c2 = (int)floorf(q2);  /* :45-49 */
  if (c0 < 0 || c0 >= res || c1 < 0 || c1 >= res || c2 < 0 || c2 >= res) return 0; /* :51-54 */
  float r0 = q0 - (float)c0, r1 = q1 - (float)c1, r2 = q2 - (float)c2;   /* :56-60 */
  int n1 = res + 1, n2 = (res + 1) * (res + 1);
  int idx[8];
  float val[8];
  idx[0] = c0 + c1 * n1 + c2 * n2;                                        /* :62-69 */
  idx[1] = c0 + c1 * n1 + (c2 + 1) * n2;
  idx[2] = c0 + (c1 + 1) * n1 + c2 * n2;
  idx[3] = c0 + (c1 + 1) * n1 + (c2 + 1) * n2;
  idx[4] = (c0 + 1) + c1 * n1 + c2 * n2;
  idx[5] = (c0 + 1) + c1 * n1 + (c2 + 1) * n2;
  idx[6] = (c0 + 1) + (c1 + 1) * n1 + c2 * n2;
  idx[7] = (c0 + 1) + (c1 + 1) * n1 + (c2 + 1) * n2;
  val[0] = ((1.0f - r0) * (1.0f - r1)) * (1.0f - r2);                     /* :71-78 */
  val[1] = ((1.0f - r0) * (1.0f - r1)) * (r2);
  val[2] = ((1.0f - r0) * (r1)) * (1.0f - r2);
  val[3] = ((1.0f - r0) * (r1)) * (r2);
  val[4] = ((r0) * (1.0f - r1)) * (1.0f - r2);
  val[5] = ((r0) * (1.0f - r1)) * (r2);
  val[6] = ((r0) * (r1)) * (1.0f - r2);
  val[7] = ((r0) * (r1)) * (r2);
  for (int a = 0; a < 3; a++) {                                           /* :83-86 left-to-right sum */
    float s = val[0] * ctr[idx[0] * 3 + a];
    for (int t = 1; t < 8; t++) s = s + val[t] * ctr[idx[t] * 3 + a];
    pos[a] = s;
  }
  return 1;
}

/* ---- A7: CIntegrateApp::Reproject pixel loop, IntegrateApp.cpp:236-268 ----
 * seg   = seg_traj_[frame_id-1]                     (row-major float64 4x4)
 * Madj  = traj[frame_id-1]^-1 * traj[0] * seg[0]^-1  (:243, computed by the caller / oracle_reproject_matrix)
 * depth is re-projected in place. */
void oracle_reproject(const oracle_volume* v, uint16_t* depth, const float* ctr, int res, float length,
                      const double* seg, const double* Madj) {
  const int n = v->cols * v->rows;
  uint16_t* buf = (uint16_t*)malloc(sizeof(uint16_t) * n);
  for (int i = 0; i < n; i++) { buf[i] = depth[i]; depth[i] = 0; }        /* :236-240 */
  float grid_ul = length / (float)res;                                    /* ControlGrid.cpp:19 */
  for (int vv = 0; vv < v->rows; vv++) {
    for (int u = 0; u < v->cols; u++) {
      uint16_t d = buf[vv * v->cols + u];
      double x, y, z;
      if (uvd2xyz(v, u, vv, d, &x, &y, &z)) {                             /* :250 */
        double q0 = ((seg[0] * x + seg[1] * y) + seg[2] * z) + seg[3] * 1.0;   /* :251 */
        double q1 = ((seg[4] * x + seg[5] * y) + seg[6] * z) + seg[7] * 1.0;
        double q2 = ((seg[8] * x + seg[9] * y) + seg[10] * z) + seg[11] * 1.0;
        float pos[3];
        if (grid_warp(ctr, res, grid_ul, (float)q0, (float)q1, (float)q2, pos)) { /* :255-256 */
          double a = (double)pos[0], b = (double)pos[1], c = (double)pos[2];
          double r0 = ((Madj[0] * a + Madj[1] * b) + Madj[2] * c) + Madj[3] * 1.0;   /* :257 */
          double r1 = ((Madj[4] * a + Madj[5] * b) + Madj[6] * c) + Madj[7] * 1.0;
          double r2 = ((Madj[8] * a + Madj[9] * b) + Madj[10] * c) + Madj[11] * 1.0;
          int uu, v2;
          uint16_t dd;
          if (xyz2uvd(v, r0, r1, r2, &uu, &v2, &dd)) {                    /* :259 */
            uint16_t ddd = depth[v2 * v->cols + uu];                      /* :260 */
            if (ddd == 0 || ddd > dd) depth[v2 * v->cols + uu] = dd;      /* :261-263 */
          }
        }
      }
    }
  }
  free(buf);
}

/* IntegrateApp.cpp:243  TiT0Ai_adj = traj[f-1].inverse() * traj[0] * seg[0].inverse() */
void oracle_reproject_matrix(const double* traj_f, const double* traj_0, const double* seg_0, double* out) {
  double a[16], b[16], t[16];
  oracle_mat4_inverse(traj_f, a);
  oracle_mat4_inverse(seg_0, b);
  mat4_mul(a, traj_0, t);
  mat4_mul(t, b, out);
}

/* IntegrateApp.cpp:71  traj[i*interval+j] = pose[i] * seg[i*interval+j] */
void oracle_compose(const double* pose, const double* seg, double* out) { mat4_mul(pose, seg, out); }

/* ---- unit access (sorted keys, like oracle/ref_driver.cpp) ---- */
int oracle_unit_count(const oracle_volume* v) { return v->n_units; }

static int cmp_int(const void* a, const void* b) { return (*(const int*)a > *(const int*)b) - (*(const int*)a < *(const int*)b); }

void oracle_unit_keys(const oracle_volume* v, int* keys) {
  for (int i = 0; i < v->n_units; i++) keys[i] = v->units[i].key;
  qsort(keys, v->n_units, sizeof(int), cmp_int);
}

int oracle_read_unit(oracle_volume* v, int key, float* sdf, float* weight) {
  oracle_unit* u = find_unit(v, key);
  if (!u) return -1;
  if (sdf) memcpy(sdf, u->sdf, sizeof(float) * UNIT_VOX);
  if (weight) memcpy(weight, u->weight, sizeof(float) * UNIT_VOX);
  return 0;
}

/* Sum of weight_ over the volume == number of voxel updates (TSDFVolume.cpp:90,94). */
double oracle_sum_weight(const oracle_volume* v) {
  double s = 0;
  for (int i = 0; i < v->n_units; i++)
    for (int l = 0; l < UNIT_VOX; l++) s += v->units[i].weight[l];
  return s;
}

/* ---- A9: TSDFVolume::SaveWorld filter, TSDFVolume.cpp:104-132.  Emits (x,y,z,intensity) float4s;
 * returns the count (call with out == NULL to size).  Unit order = ascending key (the reference's
 * unordered_map order is not canonical; compare as a set). ---- */
long oracle_extract_world(const oracle_volume* v, float* out) {
  int* keys = (int*)malloc(sizeof(int) * (v->n_units ? v->n_units : 1));
  oracle_unit_keys(v, keys);
  long n = 0;
  for (int q = 0; q < v->n_units; q++) {
    oracle_unit* unit = find_unit((oracle_volume*)v, keys[q]);
    const float* sdf = unit->sdf;
    const float* w = unit->weight;
    for (int i = 0; i < UNIT_RES; i++)
      for (int j = 0; j < UNIT_RES; j++)
        for (int k = 0; k < UNIT_RES; k++, sdf++, w++)
          if (*w != 0.0f && *sdf < 0.98f && *sdf >= -0.98f) {            /* :118 */
            if (out) {
              out[n * 4 + 0] = (float)(i + (unit->xi - 256) * 64);        /* :119-122 */
              out[n * 4 + 1] = (float)(j + (unit->yi - 256) * 64);
              out[n * 4 + 2] = (float)(k + (unit->zi - 256) * 64);
              out[n * 4 + 3] = *sdf;
            }
            n++;
          }
  }
  free(keys);
  return n;
}
