"""Path B against the COMMITTED outputs of the reference program (tests/golden/corres_golden.json, written by
tests/golden/make_golden_corres.py from oracle/_ref/BuildCorrespondence_ref and the reference's RansacCurvature.h): runs
wherever the repo is, with or without /root/reference.  CPU part: oracle/icp_oracle.cpp reproduces the reference's files."""
import hashlib

import numpy as np
import pytest

from corres_helpers import corres_golden, scene_digest, write_scene
from oracle.pyoracle import IcpOracle


def _parse(text, rows):
    """RGBDTrajectory / RGBDInformation text -> [(id1, id2, frame, matrix)] (Helper.h:21-50,73-107)."""
    lines = [l for l in text.splitlines() if l.strip()]
    out = []
    for k in range(0, len(lines), rows + 1):
        a, b, c = (int(v) for v in lines[k].split())
        out.append((a, b, c, np.array([[float(v) for v in lines[k + 1 + r].split()] for r in range(rows)])))
    return out


@pytest.fixture(scope="module")
def scene(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("corres_golden")) + "/"
    fr = write_scene(d)
    g = corres_golden()
    if scene_digest(fr) != g["scene_digest"]:
        pytest.skip("the seeded fragment generator is not bit-reproducible on this host")
    return d, fr, g


def test_restatement_reproduces_the_reference_programs_files(scene):
    d, fr, g = scene
    oc = [IcpOracle(x, n, 0.04) for x, n, _ in fr]
    init = _parse(g["init_log"], 4)
    p1, i1 = _parse(g["pass1"]["log"], 4), _parse(g["pass1"]["info"], 6)
    for (a, b, _, T0), (_, _, frame, T), (_, _, iframe, info) in zip(init, p1, i1):
        if a == 4 or b == 4:                                                     # blacklisted by black.txt
            assert frame == -1
            continue
        cnt = oc[b].count_inliers(oc[a], T0, 0.04)
        accept = cnt >= 40000 or (cnt / oc[a].n > 0.25 and cnt / oc[b].n > 0.25)
        assert accept == (frame != -1)
        if not accept:
            continue
        To, _, conv, _ = oc[b].align(oc[a], T0.astype(np.float32), max_dist=0.04)
        assert conv and np.abs(T - To.astype(np.float64)).max() <= 6e-9
        po, io_ = oc[b].find_correspondence(oc[a], To.astype(np.float64), 0.02, want_info=True)
        txt = "".join("%d %d\n" % (u, v) for u, v in po)
        assert hashlib.sha256(txt.encode()).hexdigest() == g["pass1"]["corres_sha256"]["%d_%d" % (a, b)]
        assert frame == iframe == po.shape[0] and np.allclose(info, io_, rtol=1e-12, atol=1e-7)
    # pass 2: FindCorrespondence only, from the 8-decimal transforms of refined.log
    ref2, i2 = _parse(g["pass2"]["log"], 4), _parse(g["pass2"]["info"], 6)
    for (a, b, f0, T), (_, _, frame, _), (_, _, _, info) in zip(_parse(g["refined_log"], 4), ref2, i2):
        if f0 == -1:
            assert frame == -1
            continue
        po, io_ = oc[b].find_correspondence(oc[a], T, 0.02, want_info=True)
        txt = "".join("%d %d\n" % (u, v) for u, v in po)
        assert hashlib.sha256(txt.encode()).hexdigest() == g["pass2"]["corres_sha256"]["%d_%d" % (a, b)]
        assert frame == po.shape[0] and np.allclose(info, io_, rtol=1e-12, atol=1e-7)


def test_ransac_restatement_reproduces_the_reference_headers_results(scene):
    _, fr, g = scene
    tgt, src = IcpOracle(fr[0][0], fr[0][1], 0.05), IcpOracle(fr[1][0], fr[1][1], 0.05)
    for r in g["ransac"]:
        M = np.array(r["M"], np.float32).reshape(4, 4)
        cnt, fit32, _ = src.ransac_fitness(tgt, M, r["thr"])
        ins, int_, info_s, info_t = src.ransac_inliers(tgt, M, r["thr"])
        assert cnt == r["inliers"] and np.float32(fit32).tobytes().hex() == r["fitness_f32_hex"]
        assert hashlib.sha256(ins.astype(np.int32).tobytes()).hexdigest() == r["inliers_sha256"]
        assert hashlib.sha256(int_.astype(np.int32).tobytes()).hexdigest() == r["inliers_target_sha256"]
        assert np.allclose(info_s.reshape(-1), r["info_source"], rtol=1e-13, atol=1e-9)
        assert np.allclose(info_t.reshape(-1), r["info_target"], rtol=1e-13, atol=1e-9)
