"""Generates tests/golden/tsdf_golden.json from the REFERENCE build (oracle/_ref/libref_tsdf.so =
/root/reference/Integrate/*.cpp compiled unmodified).  Run in the container that has /root/reference:

    python tests/golden/make_golden.py

The fixture holds sha256 digests of the reference's own outputs for seeded synthetic inputs
(elasticreconstruction_amd/synth.py), so the oracle restatement and the HIP path can be pinned to the
reference on machines where /root/reference does not exist (the GPU box)."""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from elasticreconstruction_amd import formats, synth  # noqa: E402
from oracle.pyoracle import RefApp  # noqa: E402


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def volume_digest(vol):
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


def rigid_case(n=6, stride=97):
    poses = synth.circle_trajectory(3000)[::stride][:n]
    depth = synth.to_numpy_u16(synth.render_depth(poses))
    return poses, depth


def warp_case(n=6, interval=3):
    return synth.make_scenario(n, interval=interval, warp=True, amplitude=0.005)


def write_run_files(sc, d):
    pose = [formats.FramedTransformation(i, i, i + 1, sc["pose"][i]) for i in range(sc["pose"].shape[0])]
    n = sc["n"]
    seg = [formats.FramedTransformation(i, i, i + 1, sc["seg"][i]) for i in range(n)]
    # one extra entry so that exactly n frames are integrated (reference off-by-one, IntegrateApp.cpp:200-203)
    formats.save_log(os.path.join(d, "pose.log"), pose + [formats.FramedTransformation(len(pose), len(pose), len(pose) + 1, sc["pose"][-1])])
    formats.save_log(os.path.join(d, "seg.log"), seg + [formats.FramedTransformation(n + j, n + j, n + j + 1, sc["seg"][-1]) for j in range(sc["interval"])])
    formats.save_ctr(os.path.join(d, "grids.ctr"), sc["grids"])


def main():
    out = {}
    here = os.path.dirname(os.path.abspath(__file__))
    poses, depth = rigid_case()
    sc = warp_case()
    # Inputs that involve LAPACK/BLAS (pose algebra) are committed, not regenerated; depth images are
    # re-rendered from them with elementwise IEEE ops only (synth.render_depth) and verified by digest.
    # The reference reads poses from .log files (8 decimals): commit exactly what it parsed.
    with tempfile.TemporaryDirectory() as d:
        write_run_files(sc, d)
        pose_l = np.stack([t.T for t in formats.load_log(os.path.join(d, "pose.log"))])[:sc["pose"].shape[0]]
        seg_l = np.stack([t.T for t in formats.load_log(os.path.join(d, "seg.log"))])[:sc["n"]]
        grids_l = formats.load_ctr(os.path.join(d, "grids.ctr"), sc["pose"].shape[0], sc["resolution"])
    from elasticreconstruction_amd.tsdf import mat4_mul
    traj_l = np.stack([mat4_mul(pose_l[f // sc["interval"]], seg_l[f]) for f in range(sc["n"])])
    np.savez(os.path.join(here, "tsdf_inputs.npz"), rigid_poses=poses, warp_world=synth.circle_trajectory(sc["n"], revolutions=1.0),
             warp_traj=traj_l, warp_pose=pose_l, warp_seg=seg_l, warp_grids=grids_l,
             warp_meta=np.array([sc["interval"], sc["resolution"]], np.int32), warp_length=np.float64(sc["length"]))
    out["rigid_depth"] = digest(depth)
    out["warp_depth"] = digest(synth.to_numpy_u16(sc["depth"]))
    ref = RefApp()
    out["scale_depth_frame0"] = digest(ref.ScaleDepth(depth[0]))
    for i in range(len(poses)):
        ref.Integrate(depth[i], poses[i])
    out["rigid"] = volume_digest(ref)
    ref.close()

    depth = synth.to_numpy_u16(sc["depth"])
    with tempfile.TemporaryDirectory() as d:
        write_run_files(sc, d)
        ref = RefApp()
        ntraj = ref.init(pose_traj=os.path.join(d, "pose.log"), seg_traj=os.path.join(d, "seg.log"),
                         ctr=os.path.join(d, "grids.ctr"), num=sc["pose"].shape[0], resolution=sc["resolution"],
                         length=sc["length"], interval=sc["interval"])
        reproj = []
        for f in range(sc["n"]):
            ex, dd, _ = ref.execute(f + 1, depth[f])
            assert ex == 0
            reproj.append(digest(dd))
        out["warp"] = volume_digest(ref)
        out["warp"]["traj_len"] = ntraj
        out["warp"]["reprojected_depth"] = reproj
        ref.close()
    with open(os.path.join(here, "tsdf_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote tsdf_golden.json:", {k: (v[:12] if isinstance(v, str) else v["sha256"][:12]) for k, v in out.items()})


if __name__ == "__main__":
    main()
