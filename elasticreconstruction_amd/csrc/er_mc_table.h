// er_mc_table.h -- the marching-cubes case table of er_tsdf_extract_mesh, GENERATED (host, once per process) instead of typed in:
// 256 cases x up to 5 triangles, each triangle three cube-edge ids, 255-terminated rows of 16 bytes.
//
// Cube conventions (shared with k_mesh and the tests):
//   corner c = cx | cy << 1 | cz << 2, (cx, cy, cz) in {0,1}^3 = offsets along the volume's i, j, k axes;
//   case index bit c is set iff the corner is INSIDE the surface (sdf < 0);
//   edge e = axis * 4 + (u | v << 1): the edge parallel to `axis` whose other two coordinates (in increasing axis order) are u, v;
//   its lower corner has coordinate 0 along `axis`.
//
// Construction (the textbook derivation of the table): on every face of the cube the inside / outside pattern of its four corners
// decides how the crossed face edges are joined -- two crossed edges: one segment; four (the ambiguous pattern, inside corners on a
// diagonal): each INSIDE corner's two face edges are joined, i.e. the outside region stays connected across the face.  The rule
// only looks at the face's own corners, so the two cubes that share a face draw the same segments on it: the surface is watertight
// across cells by construction (tests/test_mc_table.py checks exactly that for all 256 x 256 x 3 neighbour configurations).
// Every crossed edge lies on two faces, hence has two segment neighbours: the segments close into loops; each loop is oriented so
// that its normal points from the inside to the outside (towards positive sdf, free space) and fan-triangulated.
#pragma once

#include <array>
#include <cstring>
#include <vector>

namespace er {

struct McTable {
  unsigned char tri[256][16];      // edge ids, 255 = end
  unsigned char ntri[256];
};

inline void mc_edge_corners(int e, int* lo, int* hi) {
  const int axis = e >> 2, u = e & 1, v = (e >> 1) & 1;
  int c[3];
  const int o0 = axis == 0 ? 1 : 0, o1 = axis == 2 ? 1 : 2;       // the other two axes in increasing order
  c[axis] = 0; c[o0] = u; c[o1] = v;
  *lo = c[0] | c[1] << 1 | c[2] << 2;
  c[axis] = 1;
  *hi = c[0] | c[1] << 1 | c[2] << 2;
}

inline int mc_edge_between(int a, int b) {                        // the edge joining two adjacent corners
  for (int e = 0; e < 12; e++) {
    int lo, hi;
    mc_edge_corners(e, &lo, &hi);
    if ((lo == a && hi == b) || (lo == b && hi == a)) return e;
  }
  return -1;
}

inline McTable mc_table_build() {
  McTable T;
  // the six faces as corner cycles (adjacent corners consecutive)
  int face[6][4];
  int nf = 0;
  for (int axis = 0; axis < 3; axis++)
    for (int side = 0; side < 2; side++) {
      const int o0 = axis == 0 ? 1 : 0, o1 = axis == 2 ? 1 : 2;
      const int cyc[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
      for (int q = 0; q < 4; q++) {
        int c[3];
        c[axis] = side; c[o0] = cyc[q][0]; c[o1] = cyc[q][1];
        face[nf][q] = c[0] | c[1] << 1 | c[2] << 2;
      }
      nf++;
    }
  for (int cs = 0; cs < 256; cs++) {
    memset(T.tri[cs], 255, 16);
    T.ntri[cs] = 0;
    int nb[12][2], deg[12];
    for (int e = 0; e < 12; e++) deg[e] = 0;
    auto inside = [&](int c) { return ((cs >> c) & 1) != 0; };
    auto link = [&](int a, int b) { nb[a][deg[a]++] = b; nb[b][deg[b]++] = a; };
    for (int f = 0; f < 6; f++) {
      int crossed[4], nc = 0;                                     // face edge q joins face corners q and q+1
      for (int q = 0; q < 4; q++)
        if (inside(face[f][q]) != inside(face[f][(q + 1) & 3])) crossed[nc++] = q;
      if (nc == 2) {
        link(mc_edge_between(face[f][crossed[0]], face[f][(crossed[0] + 1) & 3]), mc_edge_between(face[f][crossed[1]], face[f][(crossed[1] + 1) & 3]));
      } else if (nc == 4) {
        for (int q = 0; q < 4; q++)                               // around every inside corner: its two face edges (q-1 and q)
          if (inside(face[f][q]))
            link(mc_edge_between(face[f][(q + 3) & 3], face[f][q]), mc_edge_between(face[f][q], face[f][(q + 1) & 3]));
      }
    }
    bool used[12] = {false};
    int out = 0;
    for (int e0 = 0; e0 < 12; e0++) {
      if (deg[e0] != 2 || used[e0]) continue;
      std::vector<int> loop;
      int prev = -1, cur = e0;
      do {
        loop.push_back(cur);
        used[cur] = true;
        const int nxt = nb[cur][0] != prev ? nb[cur][0] : nb[cur][1];
        prev = cur;
        cur = nxt;
      } while (cur != e0 && loop.size() < 13);
      // orientation: Newell normal of the loop (edge midpoints) against the inside -> outside direction of its edges
      double mid[12][3], N[3] = {0, 0, 0}, dir = 0;
      for (size_t q = 0; q < loop.size(); q++) {
        int lo, hi;
        mc_edge_corners(loop[q], &lo, &hi);
        for (int a = 0; a < 3; a++) mid[q][a] = 0.5 * (((lo >> a) & 1) + ((hi >> a) & 1));
      }
      for (size_t q = 0; q < loop.size(); q++) {
        const double* a = mid[q];
        const double* b = mid[(q + 1) % loop.size()];
        N[0] += (a[1] - b[1]) * (a[2] + b[2]);
        N[1] += (a[2] - b[2]) * (a[0] + b[0]);
        N[2] += (a[0] - b[0]) * (a[1] + b[1]);
      }
      for (size_t q = 0; q < loop.size(); q++) {
        int lo, hi;
        mc_edge_corners(loop[q], &lo, &hi);
        const int in = inside(lo) ? lo : hi, ot = inside(lo) ? hi : lo;
        for (int a = 0; a < 3; a++) dir += N[a] * (((ot >> a) & 1) - ((in >> a) & 1));
      }
      if (dir < 0)
        for (size_t a = 0, b = loop.size() - 1; a < b; a++, b--) std::swap(loop[a], loop[b]);
      for (size_t q = 1; q + 1 < loop.size() && out + 3 <= 15; q++) {
        T.tri[cs][out++] = (unsigned char)loop[0];
        T.tri[cs][out++] = (unsigned char)loop[q];
        T.tri[cs][out++] = (unsigned char)loop[q + 1];
      }
    }
    T.ntri[cs] = (unsigned char)(out / 3);
  }
  return T;
}

// Built once, by the initialiser of a function-local static (thread-safe since C++11: two host threads entering
// er_tsdf_extract_mesh / er_mc_table at the same time cannot both run the build -- ADVICE round 3).
inline const McTable& mc_table() {
  static const McTable T = mc_table_build();
  return T;
}

}  // namespace er
