#!/usr/bin/env python3
"""Golden vectors of the REAL PCL for path B's pin (missing 4).  This image has no PCL; the script therefore has two halves:

  python tests/golden/make_golden_pcl.py --cases DIR             (runs anywhere) writes the committed synthetic cases as PCD files + DIR/cases.txt:
        easy / hard / limit   fragment pairs of the synthetic room (test_icp_oracle.make_pair: different samplings of the same surfaces, the guess
                              0.5 / 4 / 6 degrees off; the last one is still moving when PCL's 20 iterations are used up)
        lattice               a tie-rich pair: the target is a regular 2^-8 m (3.9 mm) lattice on a plane, the source the cell centres -- four target points at
                              exactly the same float32 distance from every query, so FLANN's choice among them IS its tie rule
  python tests/golden/make_golden_pcl.py --driver ./pcl_driver   (on a machine with PCL 1.7 + FLANN; oracle/pcl_driver.cpp says how to build it)
        runs the driver on those cases and writes tests/golden/pcl_golden.json (inputs' sha256 included).

tests/test_icp_oracle.py::test_restatements_equal_real_pcl_when_its_golden_file_is_present activates itself when that file exists."""
import argparse, hashlib, json, os, subprocess, sys, tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cases():
    """[(name, (xyz0, nrm0) target, (xyz1, nrm1) source, guess 4x4 float64, reg_dist)] -- deterministic (seeded numpy only)."""
    from elasticreconstruction_amd import synth
    from test_icp_oracle import make_pair
    out = []
    for name, seed, rot, trans in (("easy", 11, 0.5, 0.005), ("hard", 21, 4.0, 0.04), ("limit", 31, 6.0, 0.06)):
        tgt, src, P = make_pair(40000, seed, rot, trans)
        out.append((name, tgt, src, np.eye(4), 0.03))
    # (dyadic coordinates: spacing 2^-8 m, centres 2^-9 off, 2^-10 above the plane -- every difference and every square is exact in float32, so the four
    #  distances are bit-identical whatever the order of the operations)
    g = np.arange(-40, 41, dtype=np.float32) * np.float32(2.0 ** -8)
    X, Y = np.meshgrid(g, g, indexing="ij")
    tgt = np.stack([X.ravel() + 1.5, Y.ravel() + 1.5, np.full(X.size, 2.0, np.float32)], 1).astype(np.float32)
    c = (g[:-1] + np.float32(2.0 ** -9)).astype(np.float32)
    Xc, Yc = np.meshgrid(c, c, indexing="ij")
    src = np.stack([Xc.ravel() + 1.5, Yc.ravel() + 1.5, np.full(Xc.size, 2.0 + 2.0 ** -10, np.float32)], 1).astype(np.float32)
    nz = lambda n: np.tile(np.array([[0, 0, -1]], np.float32), (n, 1))
    out.append(("lattice", (tgt, nz(len(tgt))), (src, nz(len(src))), np.eye(4), 0.03))
    return out


def fnv1a(idx):
    h = 1469598103934665603
    for b in np.asarray(idx, np.int32).astype("<u4").tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def write_cases(d):
    from elasticreconstruction_amd import formats
    os.makedirs(d, exist_ok=True)
    digest = {}
    with open(os.path.join(d, "cases.txt"), "w") as f:
        for name, (x0, n0), (x1, n1), G, r in cases():
            formats.save_pcd_xyzn(os.path.join(d, name + "_src.pcd"), x1, n1, binary=True)
            formats.save_pcd_xyzn(os.path.join(d, name + "_tgt.pcd"), x0, n0, binary=True)
            f.write("%s %s_src.pcd %s_tgt.pcd %.17g %s\n" % (name, name, name, r, " ".join("%.17g" % v for v in G.reshape(-1))))
            digest[name] = hashlib.sha256(x0.tobytes() + n0.tobytes() + x1.tobytes() + n1.tobytes()).hexdigest()
    return digest


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", help="only write the case directory")
    ap.add_argument("--driver", help="path of the built oracle/pcl_driver (needs PCL 1.7 + FLANN)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "pcl_golden.json"))
    a = ap.parse_args()
    if a.cases:
        print(json.dumps(write_cases(a.cases), indent=1))
        return
    if not a.driver:
        ap.error("--cases DIR or --driver PATH")
    with tempfile.TemporaryDirectory() as d:
        digest = write_cases(d)
        r = subprocess.run([a.driver, d], capture_output=True, text=True, check=True)
    g = json.loads(r.stdout)
    g["inputs_sha256"] = digest
    g["made_by"] = "tests/golden/make_golden_pcl.py --driver (oracle/pcl_driver.cpp on real PCL)"
    with open(a.out, "w") as f:
        json.dump(g, f, indent=1)
    print("wrote", a.out, "from PCL", g.get("pcl_version"))


if __name__ == "__main__":
    main()
