"""Generates tests/golden/corres_golden.json from the REFERENCE program (oracle/_ref/BuildCorrespondence_ref =
/root/reference/BuildCorrespondence/*.cpp compiled in place, unmodified, against oracle/stub_corres) and the reference's
RansacCurvature.h (oracle/_ref/libref_ransac.so).  Run in the container that has /root/reference:

    python tests/golden/make_golden_corres.py

The fixture holds the reference's own output files for the seeded scene of tests/corres_helpers.py (reg_output.log / .info as
text, sha256 of every corres_<i>_<j>.txt) plus a digest of the input clouds, so that the HIP path and the restatement stay pinned
to reference-compiled code on machines where neither /root/reference nor oracle/_ref exists.
  pass 1: --reg_traj init.log --registration --reg_dist 0.04 --output_information --blacklist black.txt
  pass 2: --reg_traj refined.log --reg_dist 0.04 --output_information   (FindCorrespondence only, on pass 1's transforms --
          byte-exact comparisons are possible here because both sides start from the same 8-decimal transforms)"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from corres_helpers import REF_BIN, ground_truth, read_outputs, run_program, scene_digest, standard_pairs, write_refined_log, write_scene  # noqa: E402
from elasticreconstruction_amd import synth  # noqa: E402
from oracle.pyoracle import RefRansac  # noqa: E402


def sha(text):
    return hashlib.sha256(text.encode()).hexdigest()


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        d = tmp + "/"
        fr = write_scene(d)
        out["scene_digest"] = scene_digest(fr)
        pairs = standard_pairs(fr, d)
        out["init_log"] = open(d + "init.log").read()
        with open(d + "black.txt", "w") as f:
            f.write("4\n")
        run_program(REF_BIN, ["--reg_traj", d + "init.log", "--registration", "--reg_dist", "0.04", "--output_information",
                              "--blacklist", d + "black.txt"], d)
        log, info, corr = read_outputs(d, pairs)
        out["pass1"] = dict(log=open(d + "reg_output.log").read(), info=open(d + "reg_output.info").read(),
                            corres_sha256={"%d_%d" % k: sha(v) for k, v in corr.items()},
                            corres_count={"%d_%d" % k: v.count("\n") for k, v in corr.items()})
        for k in corr:
            os.remove(d + "corres_%d_%d.txt" % k)
        write_refined_log(d + "refined.log", log, len(fr))
        out["refined_log"] = open(d + "refined.log").read()
        run_program(REF_BIN, ["--reg_traj", d + "refined.log", "--reg_dist", "0.04", "--output_information"], d)
        log2, info2, corr2 = read_outputs(d, pairs)
        out["pass2"] = dict(log=open(d + "reg_output.log").read(), info=open(d + "reg_output.info").read(),
                            corres_sha256={"%d_%d" % k: sha(v) for k, v in corr2.items()},
                            corres_count={"%d_%d" % k: v.count("\n") for k, v in corr2.items()})
        # RansacCurvature::getFitness / getInformation for two hypotheses of pair (0, 1)
        gt = ground_truth(fr, 0, 1)
        rs = []
        for M, thr in ((gt.astype(np.float32), 0.05), ((gt @ synth.perturbation(8, 1.0, 0.01)).astype(np.float32), 0.03)):
            ref = RefRansac(fr[1][0], fr[1][1], fr[0][0], fr[0][1], thr)
            ins, int_, fit = ref.fitness(M)
            conv, a, b, i_s, i_t = ref.align_redux(M)
            rs.append(dict(M=[float(v) for v in M.reshape(-1)], thr=thr, inliers=len(ins), fitness_f32_hex=np.float32(fit).tobytes().hex(),
                           inliers_sha256=hashlib.sha256(ins.astype(np.int32).tobytes()).hexdigest(),
                           inliers_target_sha256=hashlib.sha256(int_.astype(np.int32).tobytes()).hexdigest(),
                           info_source=[float(v) for v in i_s.reshape(-1)], info_target=[float(v) for v in i_t.reshape(-1)]))
            ref.close()
        out["ransac"] = rs
    path = os.path.join(ROOT, "tests", "golden", "corres_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
