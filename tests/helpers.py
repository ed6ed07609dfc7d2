"""Shared helpers of the parity tests."""
import hashlib
import json
import os

import numpy as np

from elasticreconstruction_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def golden():
    with open(os.path.join(GOLD, "tsdf_golden.json")) as f:
        return json.load(f)


def golden_rigid():
    """(poses [6,4,4], depth uint16 [6, 307200]) of tests/golden, depth digest verified."""
    z = np.load(os.path.join(GOLD, "tsdf_inputs.npz"))
    poses = z["rigid_poses"]
    depth = synth.to_numpy_u16(synth.render_depth(poses))
    assert digest(depth) == golden()["rigid_depth"], "synthetic renderer is not bit-reproducible on this host"
    return poses, depth


def golden_warp():
    """Scenario dict (as synth.make_scenario) rebuilt from the committed golden inputs."""
    z = np.load(os.path.join(GOLD, "tsdf_inputs.npz"))
    depth_t = synth.render_depth(z["warp_world"])
    depth = synth.to_numpy_u16(depth_t)
    assert digest(depth) == golden()["warp_depth"], "synthetic renderer is not bit-reproducible on this host"
    return dict(depth=depth_t, traj=z["warp_traj"], pose=z["warp_pose"], seg=z["warp_seg"], grids=z["warp_grids"],
                interval=int(z["warp_meta"][0]), resolution=int(z["warp_meta"][1]), length=float(z["warp_length"]),
                n=z["warp_traj"].shape[0])


def volume_digest(vol):
    """Same digest as tests/golden/make_golden.py::volume_digest for any object with unit_keys/read_unit."""
    keys = vol.unit_keys()
    h = hashlib.sha256()
    wsum = 0.0
    for k in keys:
        sdf, w = vol.read_unit(k)
        h.update(np.int32(k).tobytes())
        h.update(sdf.tobytes())
        h.update(w.tobytes())
        wsum += float(w.astype(np.float64).sum())
    return dict(keys=[int(k) for k in keys], sha256=h.hexdigest(), sum_weight=wsum)


def assert_volumes_identical(a, b, what=""):
    """Unit key sets equal and every sdf_/weight_ array bit-identical."""
    ka, kb = a.unit_keys(), b.unit_keys()
    assert np.array_equal(ka, kb), "%s unit key sets differ: %d vs %d units" % (what, len(ka), len(kb))
    for k in ka:
        sa, wa = a.read_unit(k)
        sb, wb = b.read_unit(k)
        nw = int((wa != wb).sum())
        ns = int((sa.view(np.uint32) != sb.view(np.uint32)).sum())
        assert nw == 0 and ns == 0, "%s unit %d: %d weight_ and %d sdf_ voxels differ (max |dsdf| %.3g)" % (
            what, k, nw, ns, float(np.abs(sa - sb).max()))
    return len(ka)


def oracle_run(ora, sc, depth, warp=None):
    """Frame-by-frame reference flow on the oracle: [Reproject] -> ScaleDepth -> Integrate (IntegrateApp.cpp:217-225)."""
    for f in range(depth.shape[0]):
        d = depth[f]
        if warp is not None:
            d = ora.Reproject(d, warp["ctr"][warp["grid_index"][f]], warp["resolution"], warp["length"], warp["seg"][f], warp["madj"][f])
        ora.Integrate(d, sc["traj"][f])
