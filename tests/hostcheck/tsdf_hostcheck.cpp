// tsdf_hostcheck.cpp -- TEST-ONLY host build of the per-pixel / per-voxel device functions in
// elasticreconstruction_amd/csrc/er_tsdf_math.h (ER_HD expands to plain `inline` without hipcc).
// It replays what the HIP kernels do with those functions -- frame masks per unit, frames applied in
// ascending order per voxel, scatter-min re-projection -- in plain loops, so the arithmetic the GPU will
// execute can be compared with the oracle on a machine that has no GPU.  Never shipped, never loaded by
// the product: tests/test_hostcheck.py builds it into tests/hostcheck/_build/.
#include "../../elasticreconstruction_amd/csrc/er_tsdf_math.h"

#include <algorithm>
#include <cstring>
#include <map>
#include <vector>

using namespace er;

struct HcUnit { std::vector<float> sdf, w; std::vector<int> frames; };
static int g_lo_shift = 4;             // granularity of the second-level tile_lo in the replay (k_prepare: 16-pixel tiles; 5 = no second level)
static int g_patch_shape = 1;          // 1 = the 4 x 8 x 8 box k_integrate gives a wave (default), 2 = the 8 x 8 x 8 cube of mid round 3, 0 = a 16 x 16 square of one slab
struct HcVolume { Camera cam; CameraInv cami; int cols, rows; std::map<int, HcUnit> units; long culled = 0, kept = 0, inside = 0, inside_violations = 0, sure = 0, visited = 0, unsure_pf = 0, sure_violations = 0, full_pf = 0, full_violations = 0, exact_rows = 0, exact_rows_needing = 0; };

static bool inverse4(const double* m, double* out);

extern "C" {

void* hc_create(int cols, int rows, const float* cam6) {
  HcVolume* v = new HcVolume();
  v->cam = Camera{cam6[0], cam6[1], cam6[2], cam6[3], cam6[4], cam6[5]};
  v->cami.inv_fx = 1.0 / (double)v->cam.fx; v->cami.inv_fy = 1.0 / (double)v->cam.fy;
  v->cami.pp_small = (std::fabs((double)v->cam.cx) < 1e6 && std::fabs((double)v->cam.cy) < 1e6) ? 1 : 0; v->cami.pad = 0;
  v->cols = cols; v->rows = rows;
  return v;
}
void hc_destroy(void* h) { delete static_cast<HcVolume*>(h); }

void hc_scale_depth(void* h, const uint16_t* depth, float* scaled) {
  HcVolume* v = static_cast<HcVolume*>(h);
  for (int p = 0; p < v->cols * v->rows; p++)
    scaled[p] = scale_depth_px(depth[p], scale_lambda(p % v->cols, p / v->cols, v->cam), v->cam.integration_trunc);
}

// Same contract as er_tsdf_reproject; sequential "write if empty or closer" (IntegrateApp.cpp:260-263).
void hc_reproject(void* h, uint16_t* depth, const float* ctr, int res, float length, const double* seg, const double* madj) {
  HcVolume* v = static_cast<HcVolume*>(h);
  const int n = v->cols * v->rows;
  std::vector<uint16_t> src(depth, depth + n);
  std::fill(depth, depth + n, (uint16_t)0);
  const float grid_ul = length / (float)res;
  double seg16[16] = {0};
  memcpy(seg16, seg, 12 * sizeof(double));
  cube_coord_deltas(seg16, v->cam, v->cols, v->rows, seg16 + 12);
  seg = seg16;
  for (int p = 0; p < n; p++) {
    if (src[p] == 0) continue;
    int cell; uint16_t dd;
    if (!reproject_px(p % v->cols, p / v->cols, src[p], v->cam, v->cami, v->cols, v->rows, seg, madj, ctr, res, grid_ul, cell, dd)) continue;
    if (depth[cell] == 0 || depth[cell] > dd) depth[cell] = dd;
  }
}

// Batch semantics of k_prepare + k_integrate: masks first, then per voxel all frames in order.
int hc_integrate_frames(void* h, int n, const uint16_t* depth, const double* T, const double* Tinv) {
  HcVolume* v = static_cast<HcVolume*>(h);
  const int px = v->cols * v->rows;
  std::vector<std::vector<float>> scaled(n, std::vector<float>(px));
  std::vector<FrameXform> fx(n);
  for (auto& kv : v->units) kv.second.frames.clear();
  for (int f = 0; f < n; f++) {
    const double* Tf = T + f * 16;
    const double* Ti = Tinv + f * 16;
    for (int q = 0; q < 12; q++) fx[f].mi[q] = (float)Ti[q];
    fx[f].tx = (float)Tf[3]; fx[f].ty = (float)Tf[7]; fx[f].tz = (float)Tf[11];
    for (int p = 0; p < px; p++) {
      uint16_t d = depth[(size_t)f * px + p];
      scaled[f][p] = scale_depth_px(d, scale_lambda(p % v->cols, p / v->cols, v->cam), v->cam.integration_trunc);
      if (d == 0) continue;
      int key = touch_key(p % v->cols, p / v->cols, d, v->cam, v->cami, Tf);
      if (key < 0) return -1;
      HcUnit& u = v->units[key];
      if (u.sdf.empty()) { u.sdf.assign(kUnitVox, 0.f); u.w.assign(kUnitVox, 0.f); }
      if (u.frames.empty() || u.frames.back() != f) u.frames.push_back(f);
    }
  }
  // tile maxima of the scaled depth, as k_prepare writes them (32 x 32 pixel tiles)
  const int tiles_x = (v->cols + 31) / 32, tiles_y = (v->rows + 31) / 32;
  std::vector<std::vector<float>> tile_max(n, std::vector<float>((size_t)tiles_x * tiles_y, 0.f));
  const int lts = 1 << g_lo_shift, lo_tx = (v->cols + lts - 1) / lts, lo_ty = (v->rows + lts - 1) / lts;
  std::vector<std::vector<float>> tile_lo(n, std::vector<float>((size_t)tiles_x * tiles_y, 3.0e38f));   // min over ALL pixels of the 32-pixel tile
  std::vector<std::vector<float>> tile_lo_fine(n, std::vector<float>((size_t)lo_tx * lo_ty, 3.0e38f)); // ... and of the 2^g_lo_shift-pixel tile
  for (int f = 0; f < n; f++)
    for (int p = 0; p < px; p++) {
      const size_t t = (size_t)((p / v->cols) / 32) * tiles_x + (p % v->cols) / 32;
      tile_max[f][t] = std::max(tile_max[f][t], scaled[f][p]);
      const size_t tl = (size_t)((p / v->cols) >> g_lo_shift) * lo_tx + ((p % v->cols) >> g_lo_shift);
      const float usable = scaled[f][p] > 0.001f ? scaled[f][p] : 0.0f;                          // as k_prepare: NaN / unusable -> 0
      tile_lo[f][t] = std::min(tile_lo[f][t], usable);
      tile_lo_fine[f][tl] = std::min(tile_lo_fine[f][tl], usable);
    }
  long culled = 0, kept = 0;
  for (auto& kv : v->units) {
    const int key = kv.first;
    HcUnit& u = kv.second;
    if (u.frames.empty()) continue;
    const int xi = key >> 18, yi = (key >> 9) & 511, zi = key & 511;
    const float xs = unit_shift(xi), ys = unit_shift(yi), zs = unit_shift(zi);
    // patch shape of a wave: 2 = the 8 x 8 x 8 cube (k_integrate's default), 1 = the 4 x 8 x 8 box, 0 = the 16 x 16 (j, k) square of one slab
    const int n_patches = g_patch_shape == 2 ? 8 * 64 : (g_patch_shape ? 16 * 64 : 64 * 16);
    for (int pi = 0; pi < n_patches; pi++) {
      {
        int i0, j0, k0, ni, nj, nk;
        if (g_patch_shape == 2) { i0 = (pi >> 6) * 8; j0 = ((pi >> 3) & 7) * 8; k0 = (pi & 7) * 8; ni = 8; nj = 8; nk = 8; }
        else if (g_patch_shape) { i0 = (pi >> 6) * 4; j0 = ((pi >> 3) & 7) * 8; k0 = (pi & 7) * 8; ni = 4; nj = 8; nk = 8; }
        else { i0 = pi >> 4; j0 = ((pi >> 2) & 3) * 16; k0 = (pi & 3) * 16; ni = 1; nj = 16; nk = 16; }
        // the same (patch, frame) culling k_integrate applies before its frame loop
        std::vector<int> frames;
        std::vector<char> in, ful;
        for (int f : u.frames) {
          bool inside = false, full = false;
          if (patch_may_update_box(grid_coord(i0, xs), grid_coord(i0 + ni - 1, xs), grid_coord(j0, ys), grid_coord(j0 + nj - 1, ys), grid_coord(k0, zs),
                                   grid_coord(k0 + nk - 1, zs), fx[f], v->cam, v->cols, v->rows, tile_max[f].data(), tiles_x, tiles_y, &inside,
                                   tile_lo[f].data(), &full, g_lo_shift, lo_tx, g_lo_shift < 5 ? tile_lo_fine[f].data() : nullptr)) {
            frames.push_back(f);
            in.push_back(inside);
            ful.push_back(full);
            kept++;
            v->inside += inside;
            v->full_pf += full;
          } else {
            culled++;
          }
        }
        auto voxel_of = [&](int t, int& i, int& j, int& k) {
          if (g_patch_shape) { i = i0 + (t >> 6); j = j0 + ((t >> 3) & 7); k = k0 + (t & 7); }
          else { i = i0; j = j0 + (t >> 4); k = k0 + (t & 15); }
        };
        // frame-major over the patch, like the wave of k_integrate: all lanes x register rows against frame q, then the next frame
        const int nvox = ni * nj * nk;
        float dpv[512], d2v[512];
        bool frev[512], behv[512];
        for (size_t q = 0; q < frames.size(); q++) {
          const int f = frames[q];
          bool need = false, unsure_any = false;
          bool row_need[8] = {false, false, false, false, false, false, false, false};
          for (int t = 0; t < nvox; t++) {
            int i, j, k;
            voxel_of(t, i, j, k);
            const int l = (i * 64 + j) * 64 + k;
            const float g0 = grid_coord(i, xs), g1 = grid_coord(j, ys), g2 = grid_coord(k, zs);
            float dp = 0.0f;                                // k_integrate: dp = 0 where the projection fails
            if (in[q]) {                                    // k_integrate's shortcut, cross-checked against the full test
              const unsigned pixel = voxel_project_inside(g0, g1, g2, fx[f], v->cam, v->cols, v->rows);
              unsigned ref_pixel = 0;
              const float t2 = ((fx[f].mi[8] * g0 + fx[f].mi[9] * g1) + fx[f].mi[10] * g2) + fx[f].mi[11];
              if (!voxel_project(g0, g1, g2, fx[f], v->cam, v->cols, v->rows, ref_pixel) || ref_pixel != pixel ||
                  !(t2 >= 0x1p-30f && t2 <= 0x1p30f) || pixel >= (unsigned)px) {
                v->inside_violations++;
                if (voxel_project(g0, g1, g2, fx[f], v->cam, v->cols, v->rows, ref_pixel)) dp = scaled[f][ref_pixel];
              } else {
                dp = scaled[f][pixel];
              }
            } else {
              unsigned pixel = 0;
              if (voxel_project(g0, g1, g2, fx[f], v->cam, v->cols, v->rows, pixel)) dp = scaled[f][pixel];
            }
            dpv[t] = dp;
            d2v[t] = voxel_dist2(g0, g1, g2, fx[f]);
            frev[t] = behv[t] = false;
            if (v->cam.integration_trunc < 64.0f) voxel_classify(dp, d2v[t], frev[t], behv[t]);
            const bool unsure = !(frev[t] | behv[t]);
            unsure_any |= unsure;
            need |= unsure | (frev[t] & !voxel_free_trivial(u.sdf[l], u.w[l]));
            row_need[(t >> 6) & 7] |= unsure | (frev[t] & !voxel_free_trivial(u.sdf[l], u.w[l]));
          }
          // k_integrate's "sure" path: taken for the wave when no lane needs the exact update; cross-checked lane by lane
          // against the full update (a provably-behind lane must not change, a provably-free trivial lane becomes (1, W + 1)).
          // Lanes that classify as sure in a wave that takes the exact path are checked too: the per-lane claim must hold.
          v->visited++;
          v->sure += !need;
          v->unsure_pf += unsure_any;
          if (need)
            for (int r = 0; r < nvox / 64; r++) { v->exact_rows++; v->exact_rows_needing += row_need[r]; }
          for (int t = 0; t < nvox; t++) {
            int i, j, k;
            voxel_of(t, i, j, k);
            const int l = (i * 64 + j) * 64 + k;
            float S2 = u.sdf[l], W2 = u.w[l];
            voxel_finish_d2(S2, W2, dpv[t], d2v[t]);
            if (ful[q]) {
              // k_integrate's FULL path: no projection, no sample, no arithmetic -- the update must be the tsdf = 1 update of a voxel
              // that projects inside the image onto a usable depth: (1, W + 1) for trivial voxels, (S W + 1) / (W + 1) otherwise
              const float S0 = u.sdf[l], W0 = u.w[l];
              const float S3 = voxel_free_trivial(S0, W0) ? 1.0f : (S0 * W0 + 1.0f) / (W0 + 1.0f), W3 = W0 + 1.0f;
              unsigned rp = 0;
              if (!in[q] || !voxel_project(grid_coord(i, xs), grid_coord(j, ys), grid_coord(k, zs), fx[f], v->cam, v->cols, v->rows, rp) ||
                  !(scaled[f][rp] > 0.001f) || memcmp(&S3, &S2, 4) != 0 || memcmp(&W3, &W2, 4) != 0)
                v->full_violations++;
            }
            if (behv[t] || (frev[t] && voxel_free_trivial(u.sdf[l], u.w[l]))) {
              const float S3 = frev[t] ? 1.0f : u.sdf[l], W3 = frev[t] ? u.w[l] + 1.0f : u.w[l];
              if (memcmp(&S3, &S2, 4) != 0 || memcmp(&W3, &W2, 4) != 0 || (frev[t] && behv[t])) v->sure_violations++;
            } else if (!need) {
              v->sure_violations++;                         // (cannot happen: need covers every such lane)
            }
            u.sdf[l] = S2; u.w[l] = W2;
          }
        }
      }
    }
  }
  v->culled += culled;
  v->kept += kept;
  return 0;
}

void hc_set_patch_shape(int shape) { g_patch_shape = shape; }
void hc_set_lo_shift(int shift) { g_lo_shift = shift; }
long hc_sure(void* h) { return static_cast<HcVolume*>(h)->sure; }
long hc_visited(void* h) { return static_cast<HcVolume*>(h)->visited; }
long hc_full(void* h) { return static_cast<HcVolume*>(h)->full_pf; }
long hc_full_violations(void* h) { return static_cast<HcVolume*>(h)->full_violations; }
long hc_unsure_pf(void* h) { return static_cast<HcVolume*>(h)->unsure_pf; }
long hc_exact_rows(void* h) { return static_cast<HcVolume*>(h)->exact_rows; }
long hc_exact_rows_needing(void* h) { return static_cast<HcVolume*>(h)->exact_rows_needing; }
long hc_sure_violations(void* h) { return static_cast<HcVolume*>(h)->sure_violations; }
long hc_culled(void* h) { return static_cast<HcVolume*>(h)->culled; }
long hc_inside(void* h) { return static_cast<HcVolume*>(h)->inside; }
long hc_inside_violations(void* h) { return static_cast<HcVolume*>(h)->inside_violations; }
long hc_kept(void* h) { return static_cast<HcVolume*>(h)->kept; }
int hc_unit_count(void* h) { return (int)static_cast<HcVolume*>(h)->units.size(); }
void hc_unit_keys(void* h, int* keys) { int n = 0; for (auto& kv : static_cast<HcVolume*>(h)->units) keys[n++] = kv.first; }
int hc_read_unit(void* h, int key, float* sdf, float* w) {
  HcVolume* v = static_cast<HcVolume*>(h);
  auto it = v->units.find(key);
  if (it == v->units.end()) return -1;
  memcpy(sdf, it->second.sdf.data(), kUnitVox * sizeof(float));
  memcpy(w, it->second.w.data(), kUnitVox * sizeof(float));
  return 0;
}

// Stress of the "inside" verdict of patch_may_update on its own: random cameras (focal lengths 20..2000 px, any principal
// point, images 32..1280 x 32..960), random poses up to 90 m from the origin, patches placed so that they straddle image
// borders and the camera plane as often as they sit inside.  Whenever the verdict says "inside", all 256 voxels of the patch
// are re-tested with the full voxel_project.  Returns the number of violations; *n_inside = verdicts that said "inside".
long hc_inside_stress(unsigned long long seed, long n, long* n_inside) {
  unsigned long long st = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
  auto rnd = [&]() {                                          // xorshift64*, uniform in [0, 1)
    st ^= st >> 12; st ^= st << 25; st ^= st >> 27;
    return (double)((st * 0x2545F4914F6CDD1Dull) >> 11) * (1.0 / 9007199254740992.0);
  };
  long viol = 0, ins = 0;
  for (long it = 0; it < n; it++) {
    const int cols = 32 + (int)(rnd() * 1249), rows = 32 + (int)(rnd() * 929);
    Camera cam;
    cam.fx = (float)(20.0 * pow(100.0, rnd()));
    cam.fy = (float)(20.0 * pow(100.0, rnd()));
    cam.cx = (float)(rnd() < 0.1 ? 0.0 : rnd() * cols);
    cam.cy = (float)(rnd() * rows);
    cam.icp_trunc = 2.5f; cam.integration_trunc = 2.5f;
    // random rotation from a unit quaternion, camera centre within a cube of half-width R
    double q[4], nq = 0;
    for (double& c : q) { c = rnd() * 2 - 1; nq += c * c; }
    nq = sqrt(nq) + 1e-300;
    for (double& c : q) c /= nq;
    const double Rm[9] = {1 - 2 * (q[2] * q[2] + q[3] * q[3]), 2 * (q[1] * q[2] - q[0] * q[3]), 2 * (q[1] * q[3] + q[0] * q[2]),
                          2 * (q[1] * q[2] + q[0] * q[3]), 1 - 2 * (q[1] * q[1] + q[3] * q[3]), 2 * (q[2] * q[3] - q[0] * q[1]),
                          2 * (q[1] * q[3] - q[0] * q[2]), 2 * (q[2] * q[3] + q[0] * q[1]), 1 - 2 * (q[1] * q[1] + q[2] * q[2])};
    const double Rw = rnd() < 0.5 ? 3.0 : (rnd() < 0.5 ? 30.0 : 90.0);
    const double t[3] = {(rnd() * 2 - 1) * Rw, (rnd() * 2 - 1) * Rw, (rnd() * 2 - 1) * Rw};
    FrameXform f;
    for (int r = 0; r < 3; r++) {                             // inverse pose: [R^T | -R^T t]
      for (int c = 0; c < 3; c++) f.mi[r * 4 + c] = (float)Rm[c * 3 + r];
      f.mi[r * 4 + 3] = (float)(-(Rm[0 * 3 + r] * t[0] + Rm[1 * 3 + r] * t[1] + Rm[2 * 3 + r] * t[2]));
    }
    f.tx = (float)t[0]; f.ty = (float)t[1]; f.tz = (float)t[2]; f.pad = 0.f;
    // a world point seen at pixel (pu, pv) -- anywhere from well outside to well inside the image -- at depth D
    const double pu = (rnd() * 1.6 - 0.3) * cols, pv = (rnd() * 1.6 - 0.3) * rows, D = 0.01 * pow(3000.0, rnd());
    const double pc[3] = {(pu - cam.cx) / cam.fx * D, (pv - cam.cy) / cam.fy * D, D};
    double pw[3];
    for (int r = 0; r < 3; r++) pw[r] = Rm[r * 3] * pc[0] + Rm[r * 3 + 1] * pc[1] + Rm[r * 3 + 2] * pc[2] + t[r];
    int vi[3];
    bool ok = true;
    for (int r = 0; r < 3; r++) {
      vi[r] = (int)floor(pw[r] / kUnitLength) + 256 * 64;     // voxel index in [0, 512 * 64)
      ok = ok && vi[r] >= 0 && vi[r] < 512 * 64;
    }
    if (!ok) continue;
    const float xs = unit_shift(vi[0] / 64), ys = unit_shift(vi[1] / 64), zs = unit_shift(vi[2] / 64);
    // patch shape, in turn: the 4 x 8 x 8 box k_integrate gives a wave, the 16 x 16 (j, k) square, the round-1 strip of 4 rows x 64
    const int shape = (int)(it % 3);
    const int in_ = shape == 0 ? 4 : 1, jn = shape == 0 ? 8 : (shape == 1 ? 16 : 4), kn = shape == 0 ? 8 : (shape == 1 ? 16 : 64);
    const int i0 = (vi[0] % 64) & ~(in_ - 1), j0 = (vi[1] % 64) & ~(jn - 1), k0 = (vi[2] % 64) & ~(kn - 1);
    const int tiles_x = (cols + 31) / 32, tiles_y = (rows + 31) / 32;
    std::vector<float> tile_max((size_t)tiles_x * tiles_y, 1.0e4f);   // "depth everywhere": the culling half never fires
    bool inside = false;
    if (!patch_may_update_box(grid_coord(i0, xs), grid_coord(i0 + in_ - 1, xs), grid_coord(j0, ys), grid_coord(j0 + jn - 1, ys), grid_coord(k0, zs),
                              grid_coord(k0 + kn - 1, zs), f, cam, cols, rows, tile_max.data(), tiles_x, tiles_y, &inside) || !inside)
      continue;
    ins++;
    for (int i = i0; i < i0 + in_; i++)
    for (int j = j0; j < j0 + jn; j++)
      for (int k = k0; k < k0 + kn; k++) {
        const float g0 = grid_coord(i, xs), g1 = grid_coord(j, ys), g2 = grid_coord(k, zs);
        unsigned ref_pixel = 0;
        const float t2 = ((f.mi[8] * g0 + f.mi[9] * g1) + f.mi[10] * g2) + f.mi[11];
        const unsigned pixel = voxel_project_inside(g0, g1, g2, f, cam, cols, rows);
        if (!voxel_project(g0, g1, g2, f, cam, cols, rows, ref_pixel) || ref_pixel != pixel || !(t2 >= 0x1p-30f && t2 <= 0x1p30f) ||
            pixel >= (unsigned)(cols * rows))
          viol++;
      }
  }
  if (n_inside) *n_inside = ins;
  return viol;
}

// Stress of the culling verdict (patch_may_update == false must mean that NO voxel of the patch can be updated): random
// cameras / poses / patches as above over random scaled-depth images (piecewise-constant tiles with noise, holes and values
// placed right around the patch's own distance, so that "behind the surface by more than the truncation" is decided both ways).
// Returns the number of voxels a dropped (patch, frame) pair would have updated; *n_dead = verdicts that dropped the pair.
long hc_cull_stress(unsigned long long seed, long n, long* n_dead) {
  unsigned long long st = seed * 0xD1B54A32D192ED03ull + 0x7654321ull;
  auto rnd = [&]() {
    st ^= st >> 12; st ^= st << 25; st ^= st >> 27;
    return (double)((st * 0x2545F4914F6CDD1Dull) >> 11) * (1.0 / 9007199254740992.0);
  };
  long wrong = 0, dead = 0;
  std::vector<float> img, tile_max;
  for (long it = 0; it < n; it++) {
    const int cols = 32 + (int)(rnd() * 289), rows = 32 + (int)(rnd() * 209);
    Camera cam;
    cam.fx = (float)(20.0 * pow(30.0, rnd()));
    cam.fy = (float)(20.0 * pow(30.0, rnd()));
    cam.cx = (float)(rnd() * cols);
    cam.cy = (float)(rnd() * rows);
    cam.icp_trunc = 2.5f; cam.integration_trunc = 2.5f;
    double q[4], nq = 0;
    for (double& c : q) { c = rnd() * 2 - 1; nq += c * c; }
    nq = sqrt(nq) + 1e-300;
    for (double& c : q) c /= nq;
    const double Rm[9] = {1 - 2 * (q[2] * q[2] + q[3] * q[3]), 2 * (q[1] * q[2] - q[0] * q[3]), 2 * (q[1] * q[3] + q[0] * q[2]),
                          2 * (q[1] * q[2] + q[0] * q[3]), 1 - 2 * (q[1] * q[1] + q[3] * q[3]), 2 * (q[2] * q[3] - q[0] * q[1]),
                          2 * (q[1] * q[3] - q[0] * q[2]), 2 * (q[2] * q[3] + q[0] * q[1]), 1 - 2 * (q[1] * q[1] + q[2] * q[2])};
    const double Rw = rnd() < 0.7 ? 3.0 : 40.0;
    const double t[3] = {(rnd() * 2 - 1) * Rw, (rnd() * 2 - 1) * Rw, (rnd() * 2 - 1) * Rw};
    FrameXform f;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) f.mi[r * 4 + c] = (float)Rm[c * 3 + r];
      f.mi[r * 4 + 3] = (float)(-(Rm[0 * 3 + r] * t[0] + Rm[1 * 3 + r] * t[1] + Rm[2 * 3 + r] * t[2]));
    }
    f.tx = (float)t[0]; f.ty = (float)t[1]; f.tz = (float)t[2]; f.pad = 0.f;
    const double pu = (rnd() * 1.6 - 0.3) * cols, pv = (rnd() * 1.6 - 0.3) * rows;
    const double D = (rnd() < 0.15 ? -1.0 : 1.0) * 0.02 * pow(150.0, rnd());     // some patches behind the camera
    const double pc[3] = {(pu - cam.cx) / cam.fx * D, (pv - cam.cy) / cam.fy * D, D};
    double pw[3];
    for (int r = 0; r < 3; r++) pw[r] = Rm[r * 3] * pc[0] + Rm[r * 3 + 1] * pc[1] + Rm[r * 3 + 2] * pc[2] + t[r];
    int vi[3];
    bool ok = true;
    for (int r = 0; r < 3; r++) {
      vi[r] = (int)floor(pw[r] / kUnitLength) + 256 * 64;
      ok = ok && vi[r] >= 0 && vi[r] < 512 * 64;
    }
    if (!ok) continue;
    // scaled depth image: tiles around the patch's distance from the camera (+- a few truncation widths), noise, holes
    const double dist = sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);
    const int tiles_x = (cols + 31) / 32, tiles_y = (rows + 31) / 32;
    img.assign((size_t)cols * rows, 0.f);
    tile_max.assign((size_t)tiles_x * tiles_y, 0.f);
    std::vector<float> tile_depth((size_t)tiles_x * tiles_y);
    for (float& d : tile_depth) d = rnd() < 0.2 ? 0.f : (float)(dist + (rnd() * 2 - 1) * 0.2 * (rnd() < 0.5 ? 1.0 : 5.0));
    for (int y = 0; y < rows; y++)
      for (int x = 0; x < cols; x++) {
        float d = tile_depth[(size_t)(y / 32) * tiles_x + x / 32];
        if (d > 0.f) d += (float)((rnd() * 2 - 1) * 0.01);
        if (rnd() < 0.02) d = 0.f;
        if (d < 0.f) d = 0.f;
        img[(size_t)y * cols + x] = d;
        float& m = tile_max[(size_t)(y / 32) * tiles_x + x / 32];
        m = std::max(m, d);
      }
    const float xs = unit_shift(vi[0] / 64), ys = unit_shift(vi[1] / 64), zs = unit_shift(vi[2] / 64);
    const int shape = (int)(it % 3);                           // 4 x 8 x 8 box (the default), 16 x 16 square, 4 x 64 strip (round 1)
    const int in_ = shape == 0 ? 4 : 1, jn = shape == 0 ? 8 : (shape == 1 ? 16 : 4), kn = shape == 0 ? 8 : (shape == 1 ? 16 : 64);
    const int i0 = (vi[0] % 64) & ~(in_ - 1), j0 = (vi[1] % 64) & ~(jn - 1), k0 = (vi[2] % 64) & ~(kn - 1);
    bool inside = false;
    if (patch_may_update_box(grid_coord(i0, xs), grid_coord(i0 + in_ - 1, xs), grid_coord(j0, ys), grid_coord(j0 + jn - 1, ys), grid_coord(k0, zs),
                             grid_coord(k0 + kn - 1, zs), f, cam, cols, rows, tile_max.data(), tiles_x, tiles_y, &inside))
      continue;
    dead++;
    for (int i = i0; i < i0 + in_; i++)
    for (int j = j0; j < j0 + jn; j++)
      for (int k = k0; k < k0 + kn; k++) {
        float S = 0.25f, W = 3.0f;
        if (voxel_update(S, W, grid_coord(i, xs), grid_coord(j, ys), grid_coord(k, zs), f, cam, cols, rows, img.data())) wrong++;
      }
  }
  if (n_dead) *n_dead = dead;
  return wrong;
}

// Stress of the FULL verdict of patch_may_update_box: random cameras / poses / patches (box, square, strip in turn) and depth
// images whose tiles lie around and BEHIND the patch's own distance (so that "in front of the surface by more than the
// truncation" is decided both ways), with and without holes.  Whenever the verdict says "full", every voxel of the patch is
// updated from several (S, W) states -- fresh, S == 1, arbitrary -- by the full voxel_update and by the shortcut
// ((1, W + 1) for trivial voxels, (S W + 1) / (W + 1) otherwise): the float bits must agree.  Returns the disagreements.
long hc_full_stress(unsigned long long seed, long n, long* n_full) {
  unsigned long long st = seed * 0xD1B54A32D192ED03ull + 0x13579BDFull;
  auto rnd = [&]() {
    st ^= st >> 12; st ^= st << 25; st ^= st >> 27;
    return (double)((st * 0x2545F4914F6CDD1Dull) >> 11) * (1.0 / 9007199254740992.0);
  };
  long wrong = 0, nfull = 0;
  std::vector<float> img, tile_max, tile_lo;
  for (long it = 0; it < n; it++) {
    const int cols = 64 + (int)(rnd() * 577), rows = 64 + (int)(rnd() * 417);
    Camera cam;
    cam.fx = (float)(100.0 * pow(8.0, rnd()));
    cam.fy = (float)(100.0 * pow(8.0, rnd()));
    cam.cx = (float)(rnd() * cols);
    cam.cy = (float)(rnd() * rows);
    cam.icp_trunc = 2.5f; cam.integration_trunc = 60.0f;
    double q[4], nq = 0;
    for (double& c : q) { c = rnd() * 2 - 1; nq += c * c; }
    nq = sqrt(nq) + 1e-300;
    for (double& c : q) c /= nq;
    const double Rm[9] = {1 - 2 * (q[2] * q[2] + q[3] * q[3]), 2 * (q[1] * q[2] - q[0] * q[3]), 2 * (q[1] * q[3] + q[0] * q[2]),
                          2 * (q[1] * q[2] + q[0] * q[3]), 1 - 2 * (q[1] * q[1] + q[3] * q[3]), 2 * (q[2] * q[3] - q[0] * q[1]),
                          2 * (q[1] * q[3] - q[0] * q[2]), 2 * (q[2] * q[3] + q[0] * q[1]), 1 - 2 * (q[1] * q[1] + q[2] * q[2])};
    const double Rw = rnd() < 0.7 ? 3.0 : 40.0;
    const double t[3] = {(rnd() * 2 - 1) * Rw, (rnd() * 2 - 1) * Rw, (rnd() * 2 - 1) * Rw};
    FrameXform f;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) f.mi[r * 4 + c] = (float)Rm[c * 3 + r];
      f.mi[r * 4 + 3] = (float)(-(Rm[0 * 3 + r] * t[0] + Rm[1 * 3 + r] * t[1] + Rm[2 * 3 + r] * t[2]));
    }
    f.tx = (float)t[0]; f.ty = (float)t[1]; f.tz = (float)t[2]; f.pad = 0.f;
    const double pu = (0.1 + 0.8 * rnd()) * cols, pv = (0.1 + 0.8 * rnd()) * rows;     // mostly well inside the image
    const double D = 0.3 * pow(30.0, rnd());
    const double pc[3] = {(pu - cam.cx) / cam.fx * D, (pv - cam.cy) / cam.fy * D, D};
    double pw[3];
    for (int r = 0; r < 3; r++) pw[r] = Rm[r * 3] * pc[0] + Rm[r * 3 + 1] * pc[1] + Rm[r * 3 + 2] * pc[2] + t[r];
    int vi[3];
    bool ok = true;
    for (int r = 0; r < 3; r++) {
      vi[r] = (int)floor(pw[r] / kUnitLength) + 256 * 64;
      ok = ok && vi[r] >= 0 && vi[r] < 512 * 64;
    }
    if (!ok) continue;
    const double dist = sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);
    const int tiles_x = (cols + 31) / 32, tiles_y = (rows + 31) / 32;
    img.assign((size_t)cols * rows, 0.f);
    tile_max.assign((size_t)tiles_x * tiles_y, 0.f);
    tile_lo.assign((size_t)tiles_x * tiles_y, 3.0e38f);
    // the surface: behind the patch by 0 .. 0.6 m (the verdict needs > trunc + the patch's own extent), tile to tile +- 5 cm,
    // pixel noise +- 5 mm; holes in one image out of three
    const double behind = rnd() * 0.6, hole_p = (it % 3 == 0) ? 0.002 : 0.0;
    std::vector<float> tile_depth((size_t)tiles_x * tiles_y);
    for (float& d : tile_depth) d = (float)(dist + behind + (rnd() * 2 - 1) * 0.05);
    for (int y = 0; y < rows; y++)
      for (int x = 0; x < cols; x++) {
        float d = tile_depth[(size_t)(y / 32) * tiles_x + x / 32] + (float)((rnd() * 2 - 1) * 0.005);
        if (rnd() < hole_p || d < 0.f) d = 0.f;
        img[(size_t)y * cols + x] = d;
        const size_t tt = (size_t)(y / 32) * tiles_x + x / 32;
        tile_max[tt] = std::max(tile_max[tt], d);
        tile_lo[tt] = std::min(tile_lo[tt], d > 0.001f ? d : 0.0f);
      }
    const float xs = unit_shift(vi[0] / 64), ys = unit_shift(vi[1] / 64), zs = unit_shift(vi[2] / 64);
    const int shape = (int)((it / 3) % 3);                     // 4 x 8 x 8 box (the default), 16 x 16 square, 4 x 64 strip (round 1)
    const int in_ = shape == 0 ? 4 : 1, jn = shape == 0 ? 8 : (shape == 1 ? 16 : 4), kn = shape == 0 ? 8 : (shape == 1 ? 16 : 64);
    const int i0 = (vi[0] % 64) & ~(in_ - 1), j0 = (vi[1] % 64) & ~(jn - 1), k0 = (vi[2] % 64) & ~(kn - 1);
    bool inside = false, full = false;
    if (!patch_may_update_box(grid_coord(i0, xs), grid_coord(i0 + in_ - 1, xs), grid_coord(j0, ys), grid_coord(j0 + jn - 1, ys), grid_coord(k0, zs),
                              grid_coord(k0 + kn - 1, zs), f, cam, cols, rows, tile_max.data(), tiles_x, tiles_y, &inside, tile_lo.data(), &full) ||
        !full)
      continue;
    nfull++;
    const float states[5][2] = {{0.f, 0.f}, {1.f, 7.f}, {1.f, 16777216.f}, {0.25f, 3.f}, {-0.6f, 1000.f}};
    for (int i = i0; i < i0 + in_; i++)
      for (int j = j0; j < j0 + jn; j++)
        for (int k = k0; k < k0 + kn; k++)
          for (int qq = 0; qq < 5; qq++) {
            const float S0 = states[qq][0], W0 = states[qq][1];
            float S2 = S0, W2 = W0;
            const bool upd = voxel_update(S2, W2, grid_coord(i, xs), grid_coord(j, ys), grid_coord(k, zs), f, cam, cols, rows, img.data());
            const float S3 = voxel_free_trivial(S0, W0) ? 1.0f : (S0 * W0 + 1.0f) / (W0 + 1.0f), W3 = W0 + 1.0f;
            if (!upd || memcmp(&S3, &S2, 4) != 0 || memcmp(&W3, &W2, 4) != 0) wrong++;
          }
  }
  if (n_full) *n_full = nfull;
  return wrong;
}

// Stress of voxel_classify (the square-root-free "sure" path of k_integrate) on its own: scaled depths dp from 1 mm up to the
// 64 m the shortcut is enabled for (and the special values 0, 0.001f and its neighbours, c and its neighbours), squared
// distances d2 placed ON and within a few ulps of the two decision thresholds fl(a|a|) and fl(b b), at dist = dp +- trunc
// exactly, and anywhere; voxel states fresh, S == 1, arbitrary.  Whenever a lane classifies as sure (behind, or free and
// trivial) the shortcut result must equal voxel_finish_d2's bit for bit.  Returns the violations; *n_sure = sure cases.
long hc_sure_stress(unsigned long long seed, long n, long* n_sure) {
  unsigned long long st = seed * 0xD1B54A32D192ED03ull + 0x7654321ull;
  auto rnd = [&]() {
    st ^= st >> 12; st ^= st << 25; st ^= st >> 27;
    return (double)((st * 0x2545F4914F6CDD1Dull) >> 11) * (1.0 / 9007199254740992.0);
  };
  auto bump = [](float x, int ulps) {
    for (int q = 0; q < (ulps < 0 ? -ulps : ulps); q++) x = nextafterf(x, ulps < 0 ? -3.0e38f : 3.0e38f);
    return x;
  };
  const float special[] = {0.0f, 0.001f, bump(0.001f, 1), bump(0.001f, -1), kSureBand, bump(kSureBand, 1), bump(kSureBand, -1),
                           0.03f, 0.0602f, 63.999996f, 1e-30f};
  const float states[5][2] = {{0.f, 0.f}, {1.f, 7.f}, {1.f, 16777216.f}, {0.25f, 3.f}, {-0.6f, 1000.f}};
  long wrong = 0, sure = 0;
  for (long it = 0; it < n; it++) {
    float dp = rnd() < 0.1 ? special[(int)(rnd() * 11) % 11] : (float)(0.001 * pow(64000.0, rnd()));
    if (!(dp < 64.0f)) dp = 63.999996f;
    const float a = dp - kSureBand, b = dp + kSureBand;
    float d2;
    const double pick = rnd();
    if (pick < 0.3) d2 = bump(a * fabsf(a), (int)(rnd() * 9) - 4);
    else if (pick < 0.6) d2 = bump(b * b, (int)(rnd() * 9) - 4);
    else if (pick < 0.7) { const double d = (double)dp - 0.03 + (rnd() * 2 - 1) * 2e-4; d2 = (float)(d * fabs(d)); }
    else if (pick < 0.8) { const double d = (double)dp + 0.03 + (rnd() * 2 - 1) * 2e-4; d2 = (float)(d * d); }
    else { const double d = (double)dp * (0.2 + 1.6 * rnd()); d2 = (float)(d * d); }
    if (d2 < 0.0f) d2 = 0.0f;                                        // voxel_dist2 is a sum of squares
    bool fre = false, beh = false;
    voxel_classify(dp, d2, fre, beh);
    for (int q = 0; q < 5; q++) {
      float S = states[q][0], W = states[q][1], S2 = S, W2 = W;
      voxel_finish_d2(S2, W2, dp, d2);
      if (!(beh || (fre && voxel_free_trivial(S, W)))) continue;
      sure++;
      const float S3 = fre ? 1.0f : S, W3 = fre ? W + 1.0f : W;
      if (memcmp(&S3, &S2, 4) != 0 || memcmp(&W3, &W2, 4) != 0 || (fre && beh)) wrong++;
    }
  }
  if (n_sure) *n_sure = sure;
  return wrong;
}

// touch_key (division-free, guarded) against touch_key_exact (the reference's expression): random cameras and poses with the
// camera up to 150 m from the origin (so that keys outside the 512^3 unit lattice, i.e. -1, occur), depths over the whole
// 16-bit range, and poses shifted so that the pixel's point lands within a few float64 ulps of a unit boundary (where the
// guard must hand over to the exact path).  Returns the number of disagreements; *n_out = pixels whose key is -1.
long hc_touch_key_stress(unsigned long long seed, long n, long* n_out) {
  unsigned long long st = seed * 0xA24BAED4963EE407ull + 0x9E3779B9ull;
  auto rnd = [&]() {
    st ^= st >> 12; st ^= st << 25; st ^= st >> 27;
    return (double)((st * 0x2545F4914F6CDD1Dull) >> 11) * (1.0 / 9007199254740992.0);
  };
  long wrong = 0, out = 0;
  for (long it = 0; it < n; it++) {
    Camera cam;
    cam.fx = (float)(20.0 * pow(100.0, rnd())); cam.fy = (float)(20.0 * pow(100.0, rnd()));
    cam.cx = (float)(rnd() * 1280); cam.cy = (float)(rnd() * 960);
    cam.icp_trunc = 2.5f; cam.integration_trunc = 2.5f;
    CameraInv ci;
    ci.inv_fx = 1.0 / (double)cam.fx; ci.inv_fy = 1.0 / (double)cam.fy; ci.pp_small = 1; ci.pad = 0;
    double q[4], nq = 0;
    for (double& c : q) { c = rnd() * 2 - 1; nq += c * c; }
    nq = sqrt(nq) + 1e-300;
    for (double& c : q) c /= nq;
    double T[12] = {1 - 2 * (q[2] * q[2] + q[3] * q[3]), 2 * (q[1] * q[2] - q[0] * q[3]), 2 * (q[1] * q[3] + q[0] * q[2]), 0,
                    2 * (q[1] * q[2] + q[0] * q[3]), 1 - 2 * (q[1] * q[1] + q[3] * q[3]), 2 * (q[2] * q[3] - q[0] * q[1]), 0,
                    2 * (q[1] * q[3] - q[0] * q[2]), 2 * (q[2] * q[3] + q[0] * q[1]), 1 - 2 * (q[1] * q[1] + q[2] * q[2]), 0};
    const double Rw = rnd() < 0.6 ? 3.0 : (rnd() < 0.5 ? 90.0 : 150.0);
    for (int r = 0; r < 3; r++) T[4 * r + 3] = (rnd() * 2 - 1) * Rw;
    const int u = (int)(rnd() * 1280), v = (int)(rnd() * 960);
    const uint16_t d = (uint16_t)(1 + (int)(rnd() * 65534.99));
    if (rnd() < 0.5) {                                         // move one axis onto a unit boundary (+- a few ulps)
      double x, y, z;
      uvd2xyz(u, v, d, cam, x, y, z);
      const int r = (int)(rnd() * 3) % 3;
      const double p = ((T[4 * r] * x + T[4 * r + 1] * y) + T[4 * r + 2] * z) + T[4 * r + 3];
      const double unit = floor(p / kUnitLength / 64.0 + rnd() * 2 - 1);
      double target = (unit * 64.0 - 0.5) * kUnitLength;       // voxel index flips here: p / ul + 0.5 == 64 unit
      for (int b = (int)(rnd() * 7) - 3; b != 0; b += b < 0 ? 1 : -1) target = nextafter(target, b < 0 ? -1e300 : 1e300);
      T[4 * r + 3] += target - p;
    }
    const int a = touch_key(u, v, d, cam, ci, T), b = touch_key_exact(u, v, d, cam, T);
    wrong += a != b;
    out += b < 0;
  }
  if (n_out) *n_out = out;
  return wrong;
}

// band_quotient_core against the IEEE division it replaces on the device, for every float whose magnitude bits lie in
// [lo, hi], both signs, compared as float64 bit patterns.  Returns the number of mismatches, the first one in *first.
long hc_band_quotient_check(unsigned lo, unsigned hi, unsigned* first) {
  long bad = 0;
  for (unsigned long long u = lo; u <= hi; u++)
    for (unsigned s = 0; s < 2; s++) {
      const unsigned bits = (unsigned)u | (s << 31);
      float x;
      memcpy(&x, &bits, 4);
      const double a = (double)x / kTsdfTrunc, b = band_quotient_core(x);
      if (memcmp(&a, &b, 8) != 0 && bad++ == 0 && first) *first = bits;
    }
  return bad;
}

// div1000_core against x / 1000.f for every float with bits in [lo, hi] (positive floats, +inf and the positive NaNs).
long hc_div1000_check(unsigned lo, unsigned hi, unsigned* first) {
  long bad = 0;
  for (unsigned long long u = lo; u <= hi; u++) {
    const unsigned bits = (unsigned)u;
    float x;
    memcpy(&x, &bits, 4);
    const float a = x / 1000.f, b = div1000_core(x);
    const bool same = (a != a) ? (b != b) : memcmp(&a, &b, 4) == 0;       // any NaN equals any NaN ("res > trunc" is false for all)
    if (!same && bad++ == 0 && first) *first = bits;
  }
  return bad;
}

}  // extern "C"
