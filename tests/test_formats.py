"""Round trips of every file format the two programs read or write."""
import numpy as np

from elasticreconstruction_amd import formats, synth


def test_log_info_ctr_camera_corres_roundtrip(tmp_path):
    T = [formats.FramedTransformation(i, i + 1, 7 if i else -1, synth.perturbation(i, 30, 1.0)) for i in range(4)]
    p = str(tmp_path / "a.log")
    formats.save_log(p, T)
    txt = open(p).read().splitlines()
    assert txt[0] == "0\t1\t-1" and len(txt) == 20 and len(txt[1].split(" ")) == 4
    back = formats.load_log(p)
    assert [(b.id1, b.id2, b.frame) for b in back] == [(0, 1, -1), (1, 2, 7), (2, 3, 7), (3, 4, 7)]
    assert max(np.abs(a.T - b.T).max() for a, b in zip(T, back)) < 5e-9
    with open(p, "w") as f:                      # '#' comment lines and tab separators (Matlab writer) are accepted
        f.write("# comment\n0\t1\t3\n" + "\n".join("\t".join("%.10f" % v for v in r) for r in T[0].T) + "\n")
    assert np.abs(formats.load_log(p)[0].T - T[0].T).max() < 1e-9
    info = [formats.FramedInformation(0, 1, 100, np.arange(36.0).reshape(6, 6) * 1.25)]
    formats.save_info(str(tmp_path / "a.info"), info)
    assert np.array_equal(formats.load_info(str(tmp_path / "a.info"))[0].info, info[0].info)
    grids = np.random.RandomState(0).rand(2, 27, 3).astype(np.float32) * 3
    formats.save_ctr(str(tmp_path / "g.ctr"), grids)
    assert np.array_equal(formats.load_ctr(str(tmp_path / "g.ctr"), 2, 2), grids)
    cam = np.array([517.3, 516.5, 318.6, 255.3, 2.5, 1.7], np.float32)
    formats.save_camera(str(tmp_path / "cam.txt"), cam)
    assert np.allclose(formats.load_camera(str(tmp_path / "cam.txt")), cam, atol=1e-5)
    assert np.array_equal(formats.load_camera(None), np.array([525, 525, 319.5, 239.5, 2.5, 2.5], np.float32))
    pairs = np.array([[5, 0], [7, 3], [9, 4]], np.int32)
    formats.save_corres(str(tmp_path / "c.txt"), pairs)
    assert open(str(tmp_path / "c.txt")).read() == "5 0\n7 3\n9 4\n"
    assert np.array_equal(formats.load_corres(str(tmp_path / "c.txt")), pairs)


def test_pcd_binary_ascii_compressed(tmp_path):
    rng = np.random.RandomState(3)
    xyz = rng.rand(500, 3).astype(np.float32)
    nrm = rng.randn(500, 3).astype(np.float32)
    nrm[7, 0] = np.nan
    for binary in (True, False):
        p = str(tmp_path / ("f%d.pcd" % binary))
        formats.save_pcd_xyzn(p, xyz, nrm, binary=binary)
        d = formats.load_pcd(p)
        assert np.allclose(d["x"], xyz[:, 0], atol=1e-6) and np.isnan(d["normal_x"][7])
        assert np.allclose(d["normal_z"][:7], nrm[:7, 2], atol=1e-5)
    world = rng.rand(100, 4).astype(np.float32)
    p = str(tmp_path / "world.pcd")
    formats.save_pcd_xyzi(p, world)
    d = formats.load_pcd(p)
    assert np.array_equal(np.stack([d["x"], d["y"], d["z"], d["intensity"]], 1), world)
    # binary_compressed: LZF literal-run encoding of the field-major payload is valid LZF
    soa = np.concatenate([xyz[:, 0], xyz[:, 1], xyz[:, 2]]).tobytes()
    comp = bytearray()
    for o in range(0, len(soa), 32):
        chunk = soa[o:o + 32]
        comp.append(len(chunk) - 1)
        comp += chunk
    hdr = ("# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 500\nHEIGHT 1\n"
           "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS 500\nDATA binary_compressed\n").encode()
    p = str(tmp_path / "c.pcd")
    with open(p, "wb") as f:
        f.write(hdr + np.array([len(comp), len(soa)], np.uint32).tobytes() + bytes(comp))
    d = formats.load_pcd(p)
    assert np.array_equal(d["y"], xyz[:, 1])
    # back-reference decoding
    assert formats.lzf_decompress(bytes([2, 97, 98, 99, (1 << 5) | 0, 2]), 6) == b"abcabc"


def test_host_program_compressed_pcd_writer(tmp_path):
    """`save_pcd_compressed` / `lzf_compress` of csrc/host/er_formats.h (sample.pcd of bin/FragmentOptimizer, OptApp.cpp:921-922)
    against this package's independent PCD reader; the stream must really use back references on compressible data."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "pcc")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(here, "cpp", "pcd_compressed_check.cpp"), "-lz", "-o", exe], check=True)
    rng = np.random.RandomState(11)
    names = ["x", "y", "z", "normal_x", "normal_y", "normal_z", "rgb", "curvature"]
    for n in (0, 1, 7, 5000):
        cols = rng.randn(len(names), n).astype(np.float32)
        cols[6:] = 0.0                                       # rgb / curvature are constant zero in sample.pcd
        if n > 100:
            cols[0, 50:90] = cols[0, 10:50]                  # an overlapping-distance repeat inside the 8 KiB window
        raw = str(tmp_path / "raw.bin")
        cols.tofile(raw)
        out = str(tmp_path / ("s%d.pcd" % n))
        subprocess.run([exe, "pcd", raw, str(n), out] + names, check=True)
        d = formats.load_pcd(out)
        for c, nm in enumerate(names):
            assert np.array_equal(d[nm], cols[c]), (n, nm)
        if n == 5000:
            assert os.path.getsize(out) < 0.8 * cols.nbytes   # the two zero columns (25 %) collapse
    # byte-level round trips: empty, tiny, long runs (len > 264 per reference, overlapping copies), random, text
    cases = [b"", b"a", b"ab", b"abc", b"a" * 1000, b"abcabcabc" * 500, rng.bytes(70000), (b"lattice %d\n" * 3000) % tuple(range(3000)),
             bytes(9000) + b"tail"]
    for k, blob in enumerate(cases):
        src, dst = str(tmp_path / ("b%d" % k)), str(tmp_path / ("r%d" % k))
        with open(src, "wb") as f:
            f.write(blob)
        r = subprocess.run([exe, "lzf", src, dst], check=True, capture_output=True, text=True)
        assert open(dst, "rb").read() == blob
        packed = int(r.stdout)
        assert packed <= len(blob) + len(blob) // 32 + 1
        if k in (4, 5, 8):
            assert packed < len(blob) // 10


def test_python_compressed_pcd_writer_against_the_programs_reader(tmp_path):
    """formats.lzf_compress / save_pcd_compressed (pure Python) read back by this package's reader AND by the C++ programs'
    reader (csrc/host/er_formats.h load_pcd_fields); byte-level round trips of the encoder on its edge cases."""
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "pcc")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(here, "cpp", "pcd_compressed_check.cpp"), "-lz", "-o", exe], check=True)
    rng = np.random.RandomState(5)
    for blob in (b"", b"a", b"ab", b"abc", b"a" * 1000, b"abcabcabc" * 500, rng.bytes(20000), bytes(9000) + b"tail"):
        packed = formats.lzf_compress(blob)
        assert formats.lzf_decompress(packed, len(blob)) == blob
        assert len(packed) <= len(blob) + len(blob) // 32 + 1
    names = ["x", "y", "z", "normal_x", "normal_y", "normal_z", "rgb", "curvature"]
    for n in (0, 1, 2500):
        cols = {k: rng.randn(n).astype(np.float32) for k in names}
        cols["rgb"][:] = 0
        cols["curvature"][:] = 0
        p = str(tmp_path / ("py%d.pcd" % n))
        formats.save_pcd_compressed(p, cols)
        d = formats.load_pcd(p)
        assert all(np.array_equal(d[k], cols[k]) for k in names)
        raw = str(tmp_path / "cols.bin")
        r = subprocess.run([exe, "read", p, raw] + names, check=True, capture_output=True, text=True)
        assert int(r.stdout) == n
        back = np.fromfile(raw, np.float32).reshape(len(names), n)
        assert all(np.array_equal(back[i], cols[k]) for i, k in enumerate(names))


def test_host_program_depth_png_reader_and_writer(tmp_path):
    """load_png16 / save_png16 of csrc/host/er_formats.h (Integrate --depth_list) against Pillow: 16-bit grayscale (8-bit is refused),
    every PNG filter type (smooth ramps make the encoder pick Sub / Up / Average / Paeth), odd sizes; interlaced and colour
    files are refused."""
    import os
    import subprocess
    from PIL import Image
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "pcc")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(here, "cpp", "pcd_compressed_check.cpp"), "-lz", "-o", exe], check=True)
    rng = np.random.RandomState(9)
    yy, xx = np.mgrid[0:61, 0:83]
    images = [rng.randint(0, 65536, (480, 640)).astype(np.uint16),
              ((xx * 37 + yy * 101) % 65536).astype(np.uint16),                       # ramps
              (1000 + 500 * np.sin(xx / 7.0) * np.cos(yy / 5.0)).astype(np.uint16),   # smooth
              np.zeros((1, 1), np.uint16), np.full((3, 5), 65535, np.uint16)]
    for k, img in enumerate(images):
        p = str(tmp_path / ("d%d.png" % k))
        Image.fromarray(img).save(p, optimize=bool(k % 2))
        raw = str(tmp_path / "px.bin")
        r = subprocess.run([exe, "png", p, raw], check=True, capture_output=True, text=True)
        assert r.stdout.split() == [str(img.shape[1]), str(img.shape[0])]
        assert np.array_equal(np.fromfile(raw, np.uint16).reshape(img.shape), img), k
        # the writer, read back by Pillow
        img.tofile(raw)
        q = str(tmp_path / ("w%d.png" % k))
        subprocess.run([exe, "pngw", raw, str(img.shape[1]), str(img.shape[0]), q], check=True)
        assert np.array_equal(np.asarray(Image.open(q)).astype(np.uint16), img), k
    p8 = str(tmp_path / "g8.png")
    g8 = rng.randint(0, 256, (20, 31)).astype(np.uint8)
    Image.fromarray(g8).save(p8)
    raw = str(tmp_path / "px8.bin")
    # 8-bit grayscale is not a millimetre depth map: refused (the caller would read the values as 16-bit millimetres)
    assert subprocess.run([exe, "png", p8, raw], capture_output=True).returncode != 0
    # crafted headers: a short IHDR chunk, and dimensions that would ask for a gigantic allocation
    import struct
    import zlib

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)
    sig = bytes([0x89]) + b"PNG" + bytes([0x0D, 0x0A, 0x1A, 0x0A])
    for name, ihdr in (("short", struct.pack(">II", 4, 4)), ("huge", struct.pack(">IIBBBBB", 0x7FFFFFFF, 0x7FFFFFFF, 16, 0, 0, 0, 0))):
        bad = str(tmp_path / (name + ".png"))
        with open(bad, "wb") as f:
            f.write(sig + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(b"\0" * 16)) + chunk(b"IEND", b""))
        assert subprocess.run([exe, "png", bad, raw], capture_output=True).returncode != 0, name
    rgb = str(tmp_path / "rgb.png")
    Image.fromarray(rng.randint(0, 256, (8, 8, 3)).astype(np.uint8)).save(rgb)
    assert subprocess.run([exe, "png", rgb, raw], capture_output=True).returncode != 0
    with open(str(tmp_path / "junk.png"), "wb") as f:
        f.write(b"not a png at all")
    assert subprocess.run([exe, "png", str(tmp_path / "junk.png"), raw], capture_output=True).returncode != 0


def test_host_program_text_parsers_agree_with_the_python_mirror(tmp_path):
    """load_log / save_log / load_ctr / load_camera of csrc/host/er_formats.h against formats.py on the same files, including
    comment lines, a truncated last entry and a missing camera file (reference defaults, TSDFVolumeUnit.h:69)."""
    import glob
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "pcc")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(here, "cpp", "pcd_compressed_check.cpp"), "-lz", "-o", exe], check=True)
    rng = np.random.RandomState(2)
    traj = [formats.FramedTransformation(i, i + 1, 10 * i, synth.perturbation(i, 30.0, 1.0)) for i in range(7)]
    p = str(tmp_path / "t.log")
    formats.save_log(p, traj)
    with open(p) as f:
        body = f.read()
    with open(p, "w") as f:
        f.write("# a comment where a header is expected\n" + body + "7\t8\t70\n1 0 0 0\n0 1 0 0\n")     # truncated entry at the end
    out = str(tmp_path / "t2.log")
    r = subprocess.run([exe, "log", p, out], check=True, capture_output=True, text=True)
    assert int(r.stdout) == 7
    back, mine = formats.load_log(out), formats.load_log(p)
    assert len(back) == 7 and len(mine) == 7
    for a, b, c in zip(traj, back, mine):
        assert (a.id1, a.id2, a.frame) == (b.id1, b.id2, b.frame) == (c.id1, c.id2, c.frame)
        assert np.abs(a.T - b.T).max() < 2e-8 and np.abs(a.T - c.T).max() < 1e-8
    for ref_log in sorted(glob.glob("/root/reference/Matlab_Toolbox/Example/Data/**/*.log", recursive=True))[:3]:
        r = subprocess.run([exe, "log", ref_log, out], check=True, capture_output=True, text=True)
        mine = formats.load_log(ref_log)
        back = formats.load_log(out)
        assert int(r.stdout) == len(mine) == len(back) and all(np.abs(a.T - b.T).max() < 2e-8 for a, b in zip(mine, back))
    grids = rng.randn(2, 125, 3).astype(np.float32)
    pc = str(tmp_path / "g.ctr")
    formats.save_ctr(pc, grids)
    raw = str(tmp_path / "g.bin")
    subprocess.run([exe, "ctr", pc, "2", "4", raw], check=True)
    assert np.array_equal(np.fromfile(raw, np.float32).reshape(2, 125, 3), formats.load_ctr(pc, 2, 4))
    cam = np.array([517.25, 516.5, 318.625, 255.375, 2.5, 1.75], np.float32)
    pcam = str(tmp_path / "cam.txt")
    formats.save_camera(pcam, cam)
    subprocess.run([exe, "camera", pcam, raw], check=True, capture_output=True)
    assert np.array_equal(np.fromfile(raw, np.float32), cam) and np.array_equal(formats.load_camera(pcam), cam)
    subprocess.run([exe, "camera", str(tmp_path / "absent.txt"), raw], check=True, capture_output=True)
    assert np.array_equal(np.fromfile(raw, np.float32), np.array([525, 525, 319.5, 239.5, 2.5, 2.5], np.float32))
