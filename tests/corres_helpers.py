"""Shared scene + runners of the path-B parity tests against the REFERENCE program (oracle/_ref/BuildCorrespondence_ref =
/root/reference/BuildCorrespondence/*.cpp compiled in place, unmodified, against oracle/stub_corres)."""
import os
import subprocess
from collections import OrderedDict

import numpy as np

from elasticreconstruction_amd import formats, synth

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_BIN = os.path.join(ROOT, "oracle", "_ref", "BuildCorrespondence_ref")


def write_scene(d, num=5, pts=20000, seed=77, nan_every=97, radius=0.0):
    """`num` overlapping fragments as cloud_bin_<i>.pcd in directory d (trailing slash): one ascii, one binary_compressed, the rest
    binary; every nan_every-th normal_x is NaN (LoadData drops those points, CorresApp.cpp:93-97).
    Returns [(xyz, nrm, world_T_frag)] AFTER the NaN filter, i.e. what pointclouds_ holds."""
    fr = synth.fragment_set(num, pts, seed=seed, radius=radius)
    out = []
    for i, (x, n, F) in enumerate(fr):
        n = n.copy()
        n[::nan_every, 0] = np.nan
        p = d + "cloud_bin_%d.pcd" % i
        if i == 2:
            formats.save_pcd_compressed(p, OrderedDict([("x", x[:, 0]), ("y", x[:, 1]), ("z", x[:, 2]), ("normal_x", n[:, 0]),
                                                        ("normal_y", n[:, 1]), ("normal_z", n[:, 2]), ("curvature", np.zeros(len(x), np.float32))]))
        else:
            formats.save_pcd_xyzn(p, x, n, binary=(i != 1))
        keep = ~np.isnan(n[:, 0])
        out.append((x[keep], n[keep], F))
    return out


def ground_truth(fr, i, j):
    """Transform that maps fragment j's points into fragment i's frame."""
    return np.linalg.inv(fr[i][2]) @ fr[j][2]


def standard_pairs(fr, d, rot=2.0, trans=0.02):
    """init.log over five fragments: perturbed ground truth for (0,1) (1,2) (2,3) (3,4) (1,4), a hopeless guess for (0,2)."""
    num = len(fr)
    pairs = []
    for (i, j) in [(0, 1), (0, 2), (1, 2), (2, 3), (3, 4), (1, 4)]:
        T = ground_truth(fr, i, j) @ synth.perturbation(10 * i + j, rot, trans) if (i, j) != (0, 2) else synth.perturbation(2, 70, 1.2)
        pairs.append(formats.FramedTransformation(i, j, num, T))
    formats.save_log(d + "init.log", pairs)
    return pairs


def run_program(binary, args, cwd, timeout=600):
    r = subprocess.run([binary] + args, cwd=cwd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, ER_ORACLE_QUIET="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r


def read_outputs(d, pairs):
    """reg_output.log, reg_output.info (or None) and the text of every corres_<i>_<j>.txt that exists, from directory d."""
    log = formats.load_log(d + "reg_output.log")
    info = formats.load_info(d + "reg_output.info") if os.path.exists(d + "reg_output.info") else None
    corr = {}
    for t in pairs:
        p = d + "corres_%d_%d.txt" % (t.id1, t.id2)
        if os.path.exists(p):
            corr[(t.id1, t.id2)] = open(p).read()
    return log, info, corr


def scene_digest(fr):
    """sha256 over the filtered clouds of write_scene (what pointclouds_ holds): guards the seeded generator's reproducibility."""
    import hashlib
    h = hashlib.sha256()
    for x, n, _ in fr:
        h.update(np.ascontiguousarray(x, np.float32).tobytes())
        h.update(np.ascontiguousarray(n, np.float32).tobytes())
    return h.hexdigest()


def write_refined_log(path, log, num):
    """A --reg_traj input for a FindCorrespondence-only run: the transforms of `log` (a reg_output.log), frame = num for the
    pairs that survived (LoadData takes num_ from the first record, CorresApp.cpp:72) and -1 for the rejected ones."""
    assert log[0].frame != -1
    formats.save_log(path, [formats.FramedTransformation(t.id1, t.id2, num if t.frame != -1 else -1, t.T) for t in log])


def corres_golden():
    import json
    with open(os.path.join(HERE, "golden", "corres_golden.json")) as f:
        return json.load(f)


# ---- the HARD pair list: guesses up to 6 deg / 6 cm off the ground truth -- three times the configs[2]
# perturbation -- so that PCL's 20-iteration budget, the transform criterion and the iteration limit are all reached
# (BuildCorrespondence/CorresApp.cpp:295-306).  bench.py times this very list (icp.hard_set) and runs the same checker.
from elasticreconstruction_amd.synth import hard_pair_list, pair_list  # noqa: E402,F401
from oracle.refcheck import check_pairs_against_reference, select_hard  # noqa: E402,F401
