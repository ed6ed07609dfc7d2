"""Golden digests for the sampled-stream parity tests of BASELINE.json configs[3] and configs[4] (tests/test_tsdf_gpu.py:
test_config4_hashed_grid_stream_..., test_config5_100_lattice_stream_...) from the REFERENCE build (oracle/_ref/libref_tsdf.so =
/root/reference/Integrate/*.cpp compiled unmodified) driven through CIntegrateApp::Init / Execute with the true frame ids.

    python tests/golden/make_golden_configs34.py        (in the container that has /root/reference; ~1 min)

Writes configs34_golden.json: per scene the sha256 of the sampled frame ids, of the three text files the run read, of the depth
images, and the digest of the final volume (helpers.volume_digest).  The tests compare with the reference build directly when it is
present and with these digests whenever their inputs reproduce bit for bit."""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from elasticreconstruction_amd import synth  # noqa: E402
from oracle import refcheck  # noqa: E402
import helpers  # noqa: E402

SCENES = {"config4": dict(n_frames=10000, n_runs=40, run_len=10, scene=dict(radius_drift=1.5, room=(-1.5, 4.5))),
          "config5": dict(n_frames=5000, n_runs=20, run_len=10, scene={})}


class DictVolume:
    def __init__(self, units):
        self.u = units

    def unit_keys(self):
        return np.array(sorted(self.u), np.int32)

    def read_unit(self, k):
        s, w = self.u[int(k)]
        return np.ascontiguousarray(s, np.float32), np.ascontiguousarray(w, np.float32)


def main():
    out = {}
    for tag, c in SCENES.items():
        ids = refcheck.sampled_frames(c["n_frames"], 50, c["n_runs"], c["run_len"])
        sc = synth.make_scenario(c["n_frames"], interval=50, warp=True, revolutions=c["n_frames"] / 3000.0, render_frames=ids, **c["scene"])
        depth = synth.to_numpy_u16(sc["depth"])
        with tempfile.TemporaryDirectory() as d:
            units, dt, paths = refcheck.reference_volume_of_frames(sc, depth, ids, d)
            inputs = {os.path.basename(p): hashlib.sha256(open(p, "rb").read()).hexdigest() for p in paths}
        inputs["depth"] = hashlib.sha256(depth.tobytes()).hexdigest()
        vd = helpers.volume_digest(DictVolume(units))
        out[tag] = {"frames": int(len(ids)), "job_frames": c["n_frames"], "frame_ids_sha256": hashlib.sha256(ids.tobytes()).hexdigest(),
                    "inputs": inputs, "volume": vd, "reference_seconds": dt}
        print(tag, len(ids), "frames,", len(vd["keys"]), "units, sum_weight", vd["sum_weight"], "in %.0f s" % dt, flush=True)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs34_golden.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
