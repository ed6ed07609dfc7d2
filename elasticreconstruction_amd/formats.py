"""File formats of the ElasticReconstruction pipeline (the reference's real API).

Python mirror of the readers/writers the C++ host programs (csrc/host/er_formats.h) implement;
byte-compatible with the reference:

  .log   RGBDTrajectory::LoadFromFile / SaveToFile          Integrate/TSDFVolumeUnit.h:22-63, BuildCorrespondence/Helper.h:19-60
  .info  RGBDInformation::LoadFromFile / SaveToFile          BuildCorrespondence/Helper.h:74-119
  .ctr   ControlGrid::Load                                   Integrate/ControlGrid.cpp:15-34 (writer: FragmentOptimizer/OptApp.cpp:884-895)
  camera CameraParam::LoadFromFile                           Integrate/TSDFVolumeUnit.h:72-95
  corres_<i>_<j>.txt                                         BuildCorrespondence/CorresApp.cpp:175-184
  .pcd   binary / ascii PCD v0.7 (world.pcd: TSDFVolume.cpp:104-132; cloud_bin_<i>.pcd: CorresApp.cpp:88-90)
"""
import io
import os

import numpy as np


class FramedTransformation:
    """FramedTransformation (TSDFVolumeUnit.h:12-20): id1_, id2_, frame_, transformation_ (4x4 float64)."""

    __slots__ = ("id1", "id2", "frame", "T")

    def __init__(self, id1, id2, frame, T):
        self.id1, self.id2, self.frame = int(id1), int(id2), int(frame)
        self.T = np.asarray(T, dtype=np.float64).reshape(4, 4)


def _data_lines(path):
    with open(path, "r") as f:
        for line in f:
            yield line


def load_log(path):
    """RGBDTrajectory::LoadFromFile: records of 5 lines; a line whose first char is '#' is skipped only
    where a header is expected (TSDFVolumeUnit.h:34-49).  Missing file -> empty list (f == NULL)."""
    out = []
    if not os.path.exists(path):
        return out
    it = _data_lines(path)
    for line in it:
        if len(line) > 0 and line[0] != "#":
            hdr = line.split()
            try:
                ids = [int(float(x)) for x in hdr[:3]]
            except ValueError:
                continue
            if len(ids) < 3:
                continue
            rows = []
            for _ in range(4):
                r = next(it, None)
                if r is None:
                    break
                rows.append([float(x) for x in r.split()[:4]])
            if len(rows) < 4:
                break
            out.append(FramedTransformation(ids[0], ids[1], ids[2], np.array(rows, dtype=np.float64)))
    return out


def save_log(path, traj):
    """RGBDTrajectory::SaveToFile (TSDFVolumeUnit.h:52-62): "%d\\t%d\\t%d\\n" + 4 x "%.8f %.8f %.8f %.8f\\n"."""
    with open(path, "w") as f:
        for ft in traj:
            f.write("%d\t%d\t%d\n" % (ft.id1, ft.id2, ft.frame))
            for r in range(4):
                f.write("%.8f %.8f %.8f %.8f\n" % tuple(ft.T[r]))


class FramedInformation:
    """FramedInformation (Helper.h:64-72): id1_, id2_, frame_, information_ (row-major 6x6 float64)."""

    __slots__ = ("id1", "id2", "frame", "info")

    def __init__(self, id1, id2, frame, info):
        self.id1, self.id2, self.frame = int(id1), int(id2), int(frame)
        self.info = np.asarray(info, dtype=np.float64).reshape(6, 6)


def load_info(path):
    out = []
    if not os.path.exists(path):
        return out
    it = _data_lines(path)
    for line in it:
        if len(line) > 0 and line[0] != "#":
            hdr = line.split()
            if len(hdr) < 3:
                continue
            rows = []
            for _ in range(6):
                r = next(it, None)
                if r is None:
                    break
                rows.append([float(x) for x in r.split()[:6]])
            if len(rows) < 6:
                break
            out.append(FramedInformation(int(hdr[0]), int(hdr[1]), int(hdr[2]), np.array(rows)))
    return out


def save_info(path, infos):
    with open(path, "w") as f:
        for fi in infos:
            f.write("%d\t%d\t%d\n" % (fi.id1, fi.id2, fi.frame))
            for r in range(6):
                f.write("%.8f %.8f %.8f %.8f %.8f %.8f\n" % tuple(fi.info[r]))


def load_camera(path):
    """CameraParam::LoadFromFile (TSDFVolumeUnit.h:72-95): six floats, one per line.
    Returns float32[6] = fx fy cx cy ICP_trunc integration_trunc (defaults when the file is absent)."""
    cam = np.array([525.0, 525.0, 319.5, 239.5, 2.5, 2.5], dtype=np.float32)
    if path and os.path.exists(path):
        vals = []
        with open(path) as f:
            for line in f:
                if len(line) > 0 and line[0] != "#" and line.strip():
                    vals.append(np.float32(line.split()[0]))
        for i, v in enumerate(vals[:6]):
            cam[i] = v
    return cam


def save_camera(path, cam6):
    with open(path, "w") as f:
        for v in cam6:
            f.write("%.6f\n" % float(v))


def load_ctr(path, num, resolution):
    """ControlGrid::Load x num (IntegrateApp.cpp:49-55, ControlGrid.cpp:15-34): num * (res+1)^3 lines of
    "%f %f %f"; returns float32[num, (res+1)^3, 3]."""
    verts = (resolution + 1) ** 3
    out = np.zeros((num, verts, 3), dtype=np.float32)
    with open(path) as f:
        for g in range(num):
            for i in range(verts):
                line = f.readline()
                if not line:
                    return out
                if line[0] != "#":
                    p = line.split()
                    out[g, i] = [np.float32(p[0]), np.float32(p[1]), np.float32(p[2])]
    return out


def save_ctr(path, grids):
    """Writer matching FragmentOptimizer/OptApp.cpp:884-895: "%.10f %.10f %.10f\\n" per vertex."""
    g = np.asarray(grids, dtype=np.float64).reshape(-1, 3)
    with open(path, "w") as f:
        for v in g:
            f.write("%.10f %.10f %.10f\n" % (v[0], v[1], v[2]))


def save_corres(path, pairs):
    """corres_<i>_<j>.txt (CorresApp.cpp:178-183): "%d %d\\n" = index in fragment i, index in fragment j."""
    with open(path, "w") as f:
        for a, b in np.asarray(pairs).reshape(-1, 2):
            f.write("%d %d\n" % (a, b))


def load_corres(path):
    if os.path.getsize(path) == 0:
        return np.zeros((0, 2), dtype=np.int32)
    return np.loadtxt(path, dtype=np.int32).reshape(-1, 2)


# ---------------------------------------------------------------------------------- PCD v0.7
_PCD_NP = {("F", 4): np.float32, ("F", 8): np.float64, ("U", 1): np.uint8, ("U", 2): np.uint16, ("U", 4): np.uint32,
           ("I", 1): np.int8, ("I", 2): np.int16, ("I", 4): np.int32}


def lzf_decompress(src, out_len):
    """LZF decoder (format per Matlab_Toolbox/External/matpcl/lzfd.m:21-76)."""
    src = memoryview(src)
    out = bytearray(out_len)
    ip, op, n = 0, 0, len(src)
    while ip < n:
        ctrl = src[ip]
        ip += 1
        if ctrl < 32:
            ln = ctrl + 1
            out[op:op + ln] = src[ip:ip + ln]
            ip += ln
            op += ln
        else:
            ln = ctrl >> 5
            if ln == 7:
                ln += src[ip]
                ip += 1
            ref = op - ((ctrl & 0x1F) << 8) - src[ip] - 1
            ip += 1
            for _ in range(ln + 2):
                out[op] = out[ref]
                op += 1
                ref += 1
    return bytes(out[:op])


def lzf_compress(data):
    """LZF encoder (the stream lzf_decompress / PCL's binary_compressed reader take): greedy matcher over a hash of 3-byte
    strings, back references of 3..264 bytes up to 8192 bytes back, literal runs of up to 32 bytes.  Pure Python -- meant
    for the sample-sized clouds this package writes, the C++ programs have their own (csrc/host/er_formats.h)."""
    src = bytes(data)
    n = len(src)
    out = bytearray()
    last = {}
    lit = 0
    i = 0

    def flush(end):
        nonlocal lit
        while lit < end:
            run = min(32, end - lit)
            out.append(run - 1)
            out.extend(src[lit:lit + run])
            lit += run

    while i + 2 < n:
        key = src[i:i + 3]
        ref = last.get(key, -1)
        last[key] = i
        if ref >= 0 and i - ref <= 8192:
            cap = min(n - i, 264)
            ln = 3
            while ln < cap and src[ref + ln] == src[i + ln]:
                ln += 1
            flush(i)
            back, l = i - ref - 1, ln - 2
            if l < 7:
                out.append((l << 5) | (back >> 8))
            else:
                out.append((7 << 5) | (back >> 8))
                out.append(l - 7)
            out.append(back & 0xFF)
            for k in range(i + 1, min(i + ln, n - 2)):
                last[src[k:k + 3]] = k
            i += ln
            lit = i
        else:
            i += 1
    flush(n)
    return bytes(out)


def save_pcd_compressed(path, fields):
    """pcl::PCDWriter::writeBinaryCompressed layout (FragmentOptimizer's sample.pcd, OptApp.cpp:921-922): `fields` is an
    ordered mapping name -> float32 column; columns are stored one after another, LZF-compressed, behind the two sizes."""
    names = list(fields)
    cols = [np.ascontiguousarray(fields[k], np.float32).reshape(-1) for k in names]
    n = cols[0].shape[0] if cols else 0
    assert all(c.shape[0] == n for c in cols)
    raw = b"".join(c.tobytes() for c in cols)
    packed = lzf_compress(raw)
    k = len(names)
    with open(path, "wb") as f:
        f.write(("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\nWIDTH %d\nHEIGHT 1\n"
                 "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary_compressed\n"
                 % (" ".join(names), " ".join(["4"] * k), " ".join(["F"] * k), " ".join(["1"] * k), n, n)).encode("ascii"))
        f.write(np.array([len(packed), len(raw)], np.uint32).tobytes())
        f.write(packed)


def load_pcd(path):
    """Reads ascii / binary / binary_compressed PCD v0.7 with an arbitrary field list
    (format per Matlab_Toolbox/External/matpcl/loadpcd.m:33-224).  Returns dict field -> 1-D array."""
    with open(path, "rb") as f:
        raw = f.read()
    hdr = {}
    pos = 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        k, _, v = line.partition(" ")
        hdr[k.upper()] = v.split()
        if k.upper() == "DATA":
            break
    fields = hdr["FIELDS"]
    sizes = [int(x) for x in hdr["SIZE"]]
    types = hdr["TYPE"]
    counts = [int(x) for x in hdr.get("COUNT", ["1"] * len(fields))]
    npts = int(hdr["POINTS"][0]) if "POINTS" in hdr else int(hdr["WIDTH"][0]) * int(hdr["HEIGHT"][0])
    mode = hdr["DATA"][0]
    dts = [np.dtype(_PCD_NP[(t, s)]) for t, s in zip(types, sizes)]
    out = {}
    if mode == "ascii":
        txt = raw[pos:].decode("ascii", "replace").split()
        ncol = sum(counts)
        arr = np.array(txt[:npts * ncol]).reshape(npts, ncol)
        c = 0
        for name, dt, cnt in zip(fields, dts, counts):
            col = arr[:, c:c + cnt]
            out[name] = (col.astype(np.float64).astype(dt) if dt.kind != "f" else col.astype(dt)).reshape(npts, cnt).squeeze(-1) if cnt == 1 else col.astype(dt)
            c += cnt
        return out
    if mode == "binary":
        rec = np.dtype([(name if name != "_" else "_pad%d" % i, dt, (cnt,)) if cnt > 1 else (name if name != "_" else "_pad%d" % i, dt)
                        for i, (name, dt, cnt) in enumerate(zip(fields, dts, counts))])
        a = np.frombuffer(raw, dtype=rec, count=npts, offset=pos)
        for name in a.dtype.names:
            if not name.startswith("_pad"):
                out[name] = np.ascontiguousarray(a[name])
        return out
    if mode == "binary_compressed":
        csz, usz = np.frombuffer(raw, dtype=np.uint32, count=2, offset=pos)
        blob = lzf_decompress(raw[pos + 8:pos + 8 + int(csz)], int(usz))
        off = 0
        for name, dt, cnt in zip(fields, dts, counts):   # field-major (SoA) after decompression
            nb = dt.itemsize * cnt * npts
            if name != "_":
                a = np.frombuffer(blob, dtype=dt, count=npts * cnt, offset=off)
                out[name] = a.reshape(npts, cnt).squeeze(-1) if cnt == 1 else a.reshape(npts, cnt)
            off += nb
        return out
    raise ValueError("unsupported PCD DATA mode %r" % mode)


def save_pcd_xyzi(path, xyzi):
    """world.pcd exactly as pcl::io::savePCDFile( name, PointCloud<PointXYZI>, binary=true ) lays it out
    (TSDFVolume.cpp:130): FIELDS x y z intensity, float32, DATA binary."""
    a = np.ascontiguousarray(xyzi, dtype=np.float32).reshape(-1, 4)
    n = a.shape[0]
    with open(path, "wb") as f:
        f.write(("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\n"
                 "TYPE F F F F\nCOUNT 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n"
                 % (n, n)).encode("ascii"))
        f.write(a.tobytes())


def save_pcd_xyzn(path, xyz, normals, binary=True):
    """cloud_bin_<i>.pcd-style fragment: FIELDS x y z normal_x normal_y normal_z (float32)."""
    a = np.concatenate([np.asarray(xyz, np.float32).reshape(-1, 3), np.asarray(normals, np.float32).reshape(-1, 3)], axis=1)
    n = a.shape[0]
    with open(path, "wb") as f:
        f.write(("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z normal_x normal_y normal_z\n"
                 "SIZE 4 4 4 4 4 4\nTYPE F F F F F F\nCOUNT 1 1 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\n"
                 "POINTS %d\nDATA %s\n" % (n, n, "binary" if binary else "ascii")).encode("ascii"))
        if binary:
            f.write(np.ascontiguousarray(a).tobytes())
        else:
            s = io.StringIO()
            np.savetxt(s, a, fmt="%.9g")
            f.write(s.getvalue().encode("ascii"))
