"""The marching-cubes case table of er_tsdf_extract_mesh (csrc/er_mc_table.h: generated, not typed in) -- host only, no GPU.
Checked here, for all 256 cases: the vertices of a case are exactly its crossed cube edges; at most five triangles; every triangle
faces the outside; and the WATERTIGHTNESS of the construction: for every pair of cubes that share a face (all 3 x 4096 consistent
neighbour configurations) the mesh segments the two cubes draw on that face coincide, with opposite directions."""
import ctypes as C
import itertools

import numpy as np

from elasticreconstruction_amd import _ffi


def table():
    t = np.zeros(256 * 16, np.uint8)
    assert _ffi.lib().er_mc_table(t.ctypes.data_as(C.c_void_p)) == 0
    return t.reshape(256, 16)


def edge_corners(e):
    axis, u, v = e >> 2, e & 1, (e >> 1) & 1
    o0, o1 = (1 if axis == 0 else 0), (1 if axis == 2 else 2)
    c = [0, 0, 0]
    c[o0], c[o1] = u, v
    lo = c[0] | c[1] << 1 | c[2] << 2
    return lo, lo | (1 << axis)


def triangles(row):
    n = 0
    while n < 5 and row[3 * n] != 255:
        n += 1
    assert all(v == 255 for v in row[3 * n:]), row
    return [tuple(int(v) for v in row[3 * t:3 * t + 3]) for t in range(n)]


def test_vertices_are_the_crossed_edges_and_triangles_face_outside():
    T = table()
    total = 0
    for cs in range(256):
        tris = triangles(T[cs])
        total += len(tris)
        crossed = {e for e in range(12) if ((cs >> edge_corners(e)[0]) & 1) != ((cs >> edge_corners(e)[1]) & 1)}
        assert {e for t in tris for e in t} == crossed, cs
        assert len(tris) <= 5 and (len(tris) == 0) == (cs in (0, 255))
        mid = {e: (np.array([(edge_corners(e)[0] >> a) & 1 for a in range(3)]) + np.array([(edge_corners(e)[1] >> a) & 1 for a in range(3)])) / 2.0
               for e in range(12)}
        for t in tris:
            a, b, c = (mid[e] for e in t)
            nrm = np.cross(b - a, c - a)
            # inside -> outside direction of the triangle's own edges
            d = np.zeros(3)
            for e in t:
                lo, hi = edge_corners(e)
                i, o = (lo, hi) if (cs >> lo) & 1 else (hi, lo)
                d += np.array([((o >> q) & 1) - ((i >> q) & 1) for q in range(3)])
            assert np.dot(nrm, d) >= -1e-12, (cs, t)          # (degenerate slivers of a fan may be orthogonal, never reversed)
    assert total == 820                                        # the classic table's triangle total as well


def face_segments(T, cs, axis, side):
    """Directed triangle sides of case cs whose two vertices lie on cube edges of the face `axis = side`, as pairs of
    face-local edge names (axis of the edge, its two in-face coordinates), net of sides that cancel inside the cube."""
    def on_face(e):
        ea = e >> 2
        if ea == axis:
            return None
        lo, _ = edge_corners(e)
        if ((lo >> axis) & 1) != side:
            return None
        return (ea, tuple((lo >> q) & 1 for q in range(3) if q != axis and q != ea))
    net = {}
    for t in triangles(T[cs]):
        for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
            fa, fb = on_face(a), on_face(b)
            if fa is None or fb is None:
                continue
            net[(fa, fb)] = net.get((fa, fb), 0) + 1
    out = {}
    for (fa, fb), n in net.items():
        m = n - net.get((fb, fa), 0)
        if m > 0:
            out[(fa, fb)] = m
    return out


def test_neighbouring_cubes_draw_the_same_segments_on_their_common_face():
    T = table()
    checked = 0
    for axis in range(3):
        others = [q for q in range(3) if q != axis]
        for face_bits in range(16):                                            # inside / outside of the 4 shared corners
            def full(case_free, side):
                """cube case with the shared corners on `side` of `axis` = face_bits and the other four = case_free"""
                cs = 0
                for c in range(8):
                    uv = ((c >> others[0]) & 1) | (((c >> others[1]) & 1) << 1)
                    if ((c >> axis) & 1) == side:
                        cs |= ((face_bits >> uv) & 1) << c
                    else:
                        cs |= ((case_free >> uv) & 1) << c
                return cs
            for fa, fb in itertools.product(range(16), repeat=2):
                A, B = full(fa, 1), full(fb, 0)                                # A's face axis = 1 is B's face axis = 0
                sa, sb = face_segments(T, A, axis, 1), face_segments(T, B, axis, 0)
                assert {(b, a): n for (a, b), n in sa.items()} == sb, (axis, face_bits, A, B, sa, sb)
                checked += 1
    assert checked == 3 * 16 * 256
