"""Parity tests proper for path B: HIP kernels (through the C ABI) against the CPU oracle on the same seeded
fragments.  PARITY UNPINNED w.r.t. PCL (absent): the bar is HIP == oracle with
  * integers (inlier counts, iteration counts, correspondence index lists) EXACT -- the NN search is exact and
    deterministic (float32 L2_Simple distance, ties -> lower index) on both sides;
  * ICP transforms within 1e-5 per float32 4x4 entry (the 27 point-to-plane sums are float64 but the GPU
    adds them in a different order than the sequential CPU loop);
  * information matrices within 1e-9 relative (float64 sums, different order)."""
import os

import numpy as np
import pytest

from elasticreconstruction_amd import formats, synth
from elasticreconstruction_amd.icp import Cloud, CorresApp, count_inliers, find_correspondence, icp_align
from oracle.pyoracle import IcpOracle
from test_icp_oracle import make_pair

pytestmark = pytest.mark.gpu
TOL_T = 1e-5


def clouds(pair_data, cell=0.03):
    (x0, n0), (x1, n1), P = pair_data
    return Cloud(x0, n0, cell), Cloud(x1, n1, cell), IcpOracle(x0, n0, cell), IcpOracle(x1, n1, cell), P


def test_count_inliers_exact(gpu):
    tgt, src, otgt, osrc, P = clouds(make_pair())
    for T, r in ((np.eye(4), 0.03), (P, 0.03), (P, 0.01), (synth.perturbation(1, 25, 0.5), 0.03)):
        assert count_inliers(src, tgt, T, r) == osrc.count_inliers(otgt, T, r)


def test_icp_align_matches_oracle_and_ground_truth(gpu):
    tgt, src, otgt, osrc, P = clouds(make_pair(n=120000, rot=2.0, trans=0.02))
    for rule in (0, 1):
        Tg, itg, cg, fg = icp_align(src, tgt, np.eye(4, dtype=np.float32), stop_rule=rule, want_fitness=True)
        To, ito, co, fo = osrc.align(otgt, np.eye(4, dtype=np.float32), stop_rule=rule, want_fitness=True)
        assert (itg, cg) == (ito, co), "iterations/converged differ: %s vs %s" % ((itg, cg), (ito, co))
        assert np.abs(Tg - To).max() <= TOL_T, "transform differs by %.3g" % np.abs(Tg - To).max()
        assert abs(fg - fo) <= 1e-9 * max(fo, 1e-12) + 1e-15
        assert np.abs(Tg[:3, :3].astype(np.float64) - P[:3, :3]).max() < 1e-3
        assert np.abs(Tg[:3, 3].astype(np.float64) - P[:3, 3]).max() < 1e-3
    # a non-identity guess goes through the float32 pre-transform of the source
    G = (P @ synth.perturbation(9, 0.5, 0.004)).astype(np.float32)
    Tg, itg, cg, _ = icp_align(src, tgt, G)
    To, ito, co, _ = osrc.align(otgt, G)
    assert (itg, cg) == (ito, co) and np.abs(Tg - To).max() <= TOL_T


def test_find_correspondence_lists_exact_and_information(gpu):
    tgt, src, otgt, osrc, P = clouds(make_pair())
    for dist in (0.015, 0.03):
        pg, ig = find_correspondence(src, tgt, P, dist, 0.8660, want_info=True)
        po, io = osrc.find_correspondence(otgt, P, dist, 0.8660, want_info=True)
        assert pg.shape == po.shape and pg.shape[0] > 1000
        assert np.array_equal(pg, po), "%d correspondence rows differ" % int((pg != po).any(1).sum())
        assert np.array_equal(ig[:3, :3], io[:3, :3])
        assert np.allclose(ig, io, rtol=1e-9, atol=1e-6)
    pg, _ = find_correspondence(src, tgt, synth.perturbation(2, 40, 1.0), 0.015)     # no overlap -> empty list
    assert pg.shape == (0, 2)


def test_edge_cases(gpu):
    (x0, n0), (x1, n1), P = make_pair(4000)
    tgt = Cloud(x0, n0, 0.03)
    empty = Cloud(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), 0.03)
    assert count_inliers(empty, tgt, np.eye(4), 0.03) == 0
    assert count_inliers(Cloud(x1, n1, 0.03), empty, np.eye(4), 0.03) == 0
    T, it, conv, _ = icp_align(empty, tgt, np.eye(4, dtype=np.float32))
    assert it == 0 and not conv and np.array_equal(T, np.eye(4, dtype=np.float32))   # < 3 correspondences
    from elasticreconstruction_amd import _ffi
    with pytest.raises(_ffi.ErError, match="exceeds the target's grid cell"):
        count_inliers(Cloud(x1, n1, 0.03), tgt, np.eye(4), 0.05)
    # ragged size (not a multiple of the block), duplicated points (distance ties -> lower index)
    xd = np.concatenate([x0[:1001], x0[:1001]])
    nd = np.concatenate([n0[:1001], n0[:1001]])
    td, od = Cloud(xd, nd, 0.03), IcpOracle(xd, nd, 0.03)
    s, os_ = Cloud(x0[:777], n0[:777], 0.03), IcpOracle(x0[:777], n0[:777], 0.03)
    pg, _ = find_correspondence(s, td, np.eye(4), 0.015)
    po, _ = os_.find_correspondence(od, np.eye(4), 0.015)
    assert np.array_equal(pg, po) and (pg[:, 0] < 1001).all() and pg.shape[0] == 777


def test_corres_app_pipeline_matches_reference_flow(gpu, tmp_path):
    """CorresApp mirrors CCorresApp end to end on files: cloud_bin_<i>.pcd (with NaN normals to drop), a
    registration .log with a hopeless pair (rejected by the inlier pre-check) and a blacklisted fragment;
    outputs reg_output.log/.info and corres_<i>_<j>.txt must equal the oracle-driven flow."""
    import os
    d = str(tmp_path) + "/"
    frag = synth.look_at((1.5, 1.5, 1.5), (0, 0, 1)) @ np.linalg.inv(synth.basepose())
    raw, truth = [], []
    for i in range(4):
        x, n = synth.sample_fragment(frag, 160000, seed=100 + i)
        P = synth.perturbation(200 + i, 1.0, 0.01) if i else np.eye(4)
        Pi = np.linalg.inv(P)
        x, n = (x @ Pi[:3, :3].T + Pi[:3, 3]).astype(np.float32), (n @ Pi[:3, :3].T).astype(np.float32)
        n[::97, 0] = np.nan                                              # dropped by LoadData (CorresApp.cpp:94-98)
        formats.save_pcd_xyzn(d + "cloud_bin_%d.pcd" % i, x, n)
        raw.append((x, n))
        truth.append(P)
    pairs = [formats.FramedTransformation(0, 1, 4, np.linalg.inv(truth[0]) @ truth[1] @ synth.perturbation(1, 0.5, 0.005)),
             formats.FramedTransformation(0, 2, 4, synth.perturbation(2, 60, 1.5)),          # hopeless
             formats.FramedTransformation(1, 2, 4, np.linalg.inv(truth[1]) @ truth[2]),
             formats.FramedTransformation(2, 3, 4, np.linalg.inv(truth[2]) @ truth[3])]      # 3 is blacklisted
    formats.save_log(d + "init.log", pairs)
    with open(d + "black.txt", "w") as f:
        f.write("3\n")
    app = CorresApp()
    app.out_dir = d
    app.reg_dist_, app.dist_thresh_ = 0.03, 0.015
    app.LoadData(d + "init.log", -1)
    app.Blacklist(d + "black.txt")
    app.output_information_ = True
    app.Registration()
    app.FindCorrespondence()
    app.Finalize()
    out = formats.load_log(d + "reg_output.log")
    info = formats.load_info(d + "reg_output.info")
    assert [t.frame == -1 for t in out] == [False, True, False, True]
    # oracle-driven reference flow on what LoadData parsed
    loaded = formats.load_log(d + "init.log")
    oc = []
    for x, n in raw:
        keep = ~np.isnan(n[:, 0])
        oc.append(IcpOracle(x[keep], n[keep], 0.03))
    for k, t in enumerate(loaded):
        if k in (1, 3):
            continue
        cnt = oc[t.id2].count_inliers(oc[t.id1], t.T, 0.03)
        To, ito, _, _ = oc[t.id2].align(oc[t.id1], t.T.astype(np.float32))
        assert np.abs(out[k].T - To.astype(np.float64)).max() <= TOL_T + 1e-8
        # correspondences with the GPU's own (8-decimal) transform so that index lists are comparable exactly
        po, io = oc[t.id2].find_correspondence(oc[t.id1], app.corres_traj_[k].T, 0.015, want_info=True)
        pg = formats.load_corres(d + "corres_%d_%d.txt" % (t.id1, t.id2))
        assert np.array_equal(pg, po)
        assert out[k].frame == po.shape[0] and po.shape[0] >= 0.5 * cnt
        assert np.allclose(info[k].info, io, rtol=1e-9, atol=1e-3) and info[k].frame == out[k].frame
    assert not os.path.exists(d + "corres_0_2.txt") and not os.path.exists(d + "corres_2_3.txt")


def test_icp_iteration_limits_follow_pcl_do_while(gpu):
    """max_iter = 1 and max_iter = 0 both run exactly ONE iteration and report converged (PCL's do { ... } while( !converged )
    with `iterations >= max_iterations`), max_iter = 3 stops at 3; checked against the restatement, which is
    itself checked against the stub's second statement (tests/test_corres_reference.py)."""
    (x0, n0), (x1, n1), P = make_pair(n=30000, rot=4.0, trans=0.04)
    tgt, src = Cloud(x0, n0, 0.03), Cloud(x1, n1, 0.03)
    otgt, osrc = IcpOracle(x0, n0, 0.03), IcpOracle(x1, n1, 0.03)
    g = np.eye(4, dtype=np.float32)
    for max_iter, want_iter in ((1, 1), (0, 1), (-5, 1), (3, 3)):
        Tg, itg, cg, _ = icp_align(src, tgt, g, max_iter=max_iter)
        To, ito, co, _ = osrc.align(otgt, g, max_iter=max_iter)
        assert (itg, cg) == (ito, co) == (want_iter, True), (max_iter, itg, cg, ito, co)
        assert np.abs(Tg - To).max() <= TOL_T


def test_batch_entry_points_equal_single_calls(gpu):
    """er_*_batch pipelines pairs over several streams/workspaces; the results must be those of the single-pair calls
    (integers and index lists exact; transforms bit-identical: same kernels, same float64 atomics order is NOT
    guaranteed, so 1e-6), for more pairs than lanes, mixed sizes, a rejected pair and shared clouds, and from
    several host threads at once (clouds are immutable and shareable)."""
    import threading
    from elasticreconstruction_amd.icp import count_inliers_batch, find_correspondence_batch, icp_align_batch
    data = [make_pair(n=60000 + 9000 * i, rot=1.0 + 0.3 * i, trans=0.01, seed=40 + i) for i in range(3)]
    cl = [(Cloud(x0, n0, 0.03), Cloud(x1, n1, 0.03), P) for (x0, n0), (x1, n1), P in data]
    empty = Cloud(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), 0.03)
    srcs, tgts, Ts = [], [], []
    for k in range(11):                                   # > kLanes pairs, clouds reused by several pairs
        tgt, src, P = cl[k % 3]
        srcs.append(src); tgts.append(tgt)
        Ts.append(P @ synth.perturbation(80 + k, 0.8, 0.006) if k != 4 else synth.perturbation(3, 40, 1.0))
    srcs.append(empty); tgts.append(cl[0][0]); Ts.append(np.eye(4))
    cb = count_inliers_batch(srcs, tgts, Ts, 0.03)
    assert [int(c) for c in cb] == [count_inliers(s, t, T, 0.03) for s, t, T in zip(srcs, tgts, Ts)]
    assert cb[4] < 2000 and cb[-1] == 0
    Fb, itb, cvb, fitb = icp_align_batch(srcs, tgts, [T.astype(np.float32) for T in Ts], want_fitness=True)
    for k, (s, t, T) in enumerate(zip(srcs, tgts, Ts)):
        F1, it1, cv1, fit1 = icp_align(s, t, T.astype(np.float32), want_fitness=True)
        assert (int(itb[k]), bool(cvb[k])) == (it1, cv1), "pair %d" % k
        assert np.abs(Fb[k] - F1).max() <= 1e-6
        assert fitb[k] == pytest.approx(fit1, rel=1e-9) or (fitb[k] > 1e300 and fit1 > 1e300)
    lists, infos = find_correspondence_batch(srcs, tgts, [F.astype(np.float64) for F in Fb], 0.015, 0.8660, want_info=True)
    for k, (s, t) in enumerate(zip(srcs, tgts)):
        p1, i1 = find_correspondence(s, t, Fb[k].astype(np.float64), 0.015, 0.8660, want_info=True)
        assert np.array_equal(lists[k], p1), "pair %d" % k
        assert np.allclose(infos[k], i1, rtol=1e-9, atol=1e-6)
    assert lists[4].shape[0] == 0 and lists[-1].shape[0] == 0

    # concurrent callers sharing clouds
    out, errs = {}, []

    def worker(w):
        try:
            out[w] = [int(c) for c in count_inliers_batch(srcs, tgts, Ts, 0.03)], icp_align_batch(srcs[:5], tgts[:5], [T.astype(np.float32) for T in Ts[:5]])[1].tolist()
        except Exception as ex:                               # pragma: no cover
            errs.append(ex)
    th = [threading.Thread(target=worker, args=(w,)) for w in range(4)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for w in range(4):
        assert out[w][0] == [int(c) for c in cb] and out[w][1] == [int(v) for v in itb[:5]]


def test_ransac_fitness_batch_matches_oracle(gpu):
    """er_ransac_fitness_batch = RansacCurvature::getFitness over a list of hypotheses.  Inlier counts
    exact; fitness within 1e-9 of the oracle's float64 sum / count and 1e-4 of its float32 running sum (the reference's)."""
    from elasticreconstruction_amd.icp import ransac_fitness_batch
    (x0, n0), (x1, n1), P = make_pair(n=30000, rot=3.0, trans=0.03)
    keep = np.random.default_rng(3).choice(x1.shape[0], 4000, replace=False)      # the reference scores a down-sampled source
    x1s, n1s = x1[np.sort(keep)], n1[np.sort(keep)]
    tgt, src = Cloud(x0, n0, 0.05), Cloud(x1s, n1s, 0.05)
    otgt, osrc = IcpOracle(x0, n0, 0.05), IcpOracle(x1s, n1s, 0.05)
    hyps = [P.astype(np.float32), np.eye(4, dtype=np.float32), synth.perturbation(1, 60, 2.0).astype(np.float32)]
    hyps += [(P @ synth.perturbation(100 + k, 0.2 * k, 0.002 * k)).astype(np.float32) for k in range(29)]
    for thr in (0.05, 0.02):
        cnt, fit = ransac_fitness_batch(src, tgt, hyps, thr)
        for h, M in enumerate(hyps):
            c, f32, s64 = osrc.ransac_fitness(otgt, M, thr)
            assert int(cnt[h]) == c, "hypothesis %d: %d vs %d inliers" % (h, cnt[h], c)
            if c:
                assert fit[h] == pytest.approx(s64 / c, rel=1e-9)
                assert fit[h] == pytest.approx(f32, rel=1e-4)
            else:
                assert fit[h] == float(np.finfo(np.float32).max)
        assert cnt[0] > 0.3 * len(keep) and cnt[2] == 0
    # the accepted hypothesis with its lists: getFitness's inliers / inliers_target + getInformation (RansacCurvature.h:661-733)
    from elasticreconstruction_amd.icp import ransac_inliers
    for M, thr in ((hyps[0], 0.05), (hyps[7], 0.02), (hyps[2], 0.05)):
        ins, int_, f, info_s, info_t = ransac_inliers(src, tgt, M, thr)
        o_ins, o_int, o_is, o_it = osrc.ransac_inliers(otgt, M, thr)
        c, f32, s64 = osrc.ransac_fitness(otgt, M, thr)
        assert np.array_equal(ins, o_ins) and np.array_equal(int_, o_int)                 # same lists, same order
        assert np.allclose(info_s, o_is, rtol=1e-12, atol=1e-9) and np.allclose(info_t, o_it, rtol=1e-12, atol=1e-9)
        assert f == (pytest.approx(s64 / c, rel=1e-9) if c else float(np.finfo(np.float32).max))
    # more hypotheses than one launch's grid.y chunk, tiny source
    tiny = Cloud(x1s[:300], n1s[:300], 0.05)
    many = np.repeat(np.stack(hyps[:4])[None], 8200, axis=0).reshape(-1, 4, 4)
    cnt, fit = ransac_fitness_batch(tiny, tgt, many, 0.05)
    assert cnt.shape[0] == 32800 and np.array_equal(cnt[:4], cnt[-4:]) and np.array_equal(cnt.reshape(-1, 4), np.tile(cnt[:4], (8200, 1)))
    with pytest.raises(Exception):
        ransac_fitness_batch(src, tgt, hyps, 0.08)               # radius beyond the target's grid cell


def test_ransac_entry_points_equal_the_reference_header(gpu, tmp_path):
    """er_ransac_fitness_batch / er_ransac_inliers against the reference's OWN RansacCurvature.h (getFitness :661-704,
    getInformation :707-733, included in place behind oracle/_ref/libref_ransac.so when that library travelled here) and against
    its committed results (tests/golden/corres_golden.json): inlier lists identical, fitness within 1e-6 of the reference's
    float32 running sum (the GPU sums in float64), information within 1e-12."""
    import hashlib
    from corres_helpers import corres_golden, scene_digest, write_scene
    from elasticreconstruction_amd.icp import ransac_fitness_batch, ransac_inliers
    from oracle.pyoracle import RefRansac
    fr = write_scene(str(tmp_path) + "/")
    g = corres_golden()
    golden_ok = scene_digest(fr) == g["scene_digest"]
    assert golden_ok or RefRansac.available()
    tgt, src = Cloud(fr[0][0], fr[0][1], 0.05), Cloud(fr[1][0], fr[1][1], 0.05)
    for r in g["ransac"]:
        M = np.array(r["M"], np.float32).reshape(4, 4)
        cnt, fit = ransac_fitness_batch(src, tgt, [M], r["thr"])
        ins, int_, f, info_s, info_t = ransac_inliers(src, tgt, M, r["thr"])
        if RefRansac.available():
            ref = RefRansac(fr[1][0], fr[1][1], fr[0][0], fr[0][1], r["thr"])
            r_ins, r_int, r_fit = ref.fitness(M)
            conv, a, b, r_is, r_it = ref.align_redux(M)
            ref.close()
            assert int(cnt[0]) == len(r_ins) and np.array_equal(ins, r_ins) and np.array_equal(int_, r_int) and conv
            assert fit[0] == pytest.approx(r_fit, rel=1e-5) and f == pytest.approx(r_fit, rel=1e-5)
            assert np.allclose(info_s, r_is, rtol=1e-12, atol=1e-9) and np.allclose(info_t, r_it, rtol=1e-12, atol=1e-9)
        if golden_ok:
            assert int(cnt[0]) == r["inliers"]
            assert hashlib.sha256(ins.astype(np.int32).tobytes()).hexdigest() == r["inliers_sha256"]
            assert hashlib.sha256(int_.astype(np.int32).tobytes()).hexdigest() == r["inliers_target_sha256"]
            assert fit[0] == pytest.approx(float(np.frombuffer(bytes.fromhex(r["fitness_f32_hex"]), np.float32)[0]), rel=1e-5)
            assert np.allclose(info_s.reshape(-1), r["info_source"], rtol=1e-12, atol=1e-9)
            assert np.allclose(info_t.reshape(-1), r["info_target"], rtol=1e-12, atol=1e-9)


def test_config2_size_50_pairs_over_25_distinct_fragments(gpu):
    """BASELINE.json configs[2] at full size: 50 pairs over 25 DISTINCT fragments of 250 k points each, through the reference's
    two loops (Registration: pre-check + ICP; FindCorrespondence + information matrix) as bench.py runs them (the *_batch
    entry points), every pair checked against the CPU oracle: inlier counts, iteration counts, convergence flags and
    correspondence index lists EXACT, transforms within 1e-5 per float32 entry, information matrices within 1e-9 relative."""
    from elasticreconstruction_amd.icp import count_inliers_batch, find_correspondence_batch, icp_align_batch
    n_frag, n_pairs = 25, 50
    frs = synth.fragment_set(n_frag, 250000, device="cuda:0")
    assert all(len(x) == 250000 for x, _, _ in frs)
    gc = [Cloud(x, n, 0.03) for x, n, _ in frs]
    oc = [IcpOracle(x, n, 0.03) for x, n, _ in frs]
    pairs = []
    for k in range(n_pairs):
        a = k % n_frag
        b = (a + 1 + (k // n_frag) % 3) % n_frag
        pairs.append((a, b, np.linalg.inv(frs[a][2]) @ frs[b][2] @ synth.perturbation(700 + k, 2.0, 0.02)))
    assert len({q for a, b, _ in pairs for q in (a, b)}) == n_frag
    srcs, tgts = [gc[b] for _, b, _ in pairs], [gc[a] for a, _, _ in pairs]
    cnts = count_inliers_batch(srcs, tgts, [T for _, _, T in pairs], 0.03)
    fins, iters, conv, _ = icp_align_batch(srcs, tgts, [T.astype(np.float32) for _, _, T in pairs], 0.03, 20, 1e-6, 0)
    lists, infos = find_correspondence_batch(srcs, tgts, [F.astype(np.float64) for F in fins], 0.015, 0.8660, True)
    worst_T = 0.0
    for k, (a, b, T) in enumerate(pairs):
        assert int(cnts[k]) == oc[b].count_inliers(oc[a], T, 0.03), "pair %d: inlier count" % k
        To, ito, co, _ = oc[b].align(oc[a], T.astype(np.float32), 0.03, 20, 1e-6, 0)
        assert (int(iters[k]), bool(conv[k])) == (ito, co), "pair %d: iterations/converged %s vs %s" % (k, (iters[k], conv[k]), (ito, co))
        worst_T = max(worst_T, float(np.abs(fins[k] - To).max()))
        assert np.abs(fins[k] - To).max() <= TOL_T, "pair %d: transform differs by %.3g" % (k, np.abs(fins[k] - To).max())
        po, io = oc[b].find_correspondence(oc[a], fins[k].astype(np.float64), 0.015, 0.8660, want_info=True)
        assert np.array_equal(lists[k], po), "pair %d: %d vs %d correspondences" % (k, lists[k].shape[0], po.shape[0])
        assert np.allclose(infos[k], io, rtol=1e-9, atol=1e-6)
        # the ICP recovered the ground truth from the <= 2 deg / 2 cm perturbation
        gt = np.linalg.inv(frs[a][2]) @ frs[b][2]
        assert np.abs(fins[k].astype(np.float64) - gt).max() < 2e-3, "pair %d: ground truth missed" % k
    assert int(np.sum(iters)) >= n_pairs and min(l.shape[0] for l in lists) > 50000
    print("configs[2]: 50 pairs, mean %.2f ICP iterations, max |T_gpu - T_oracle| = %.2g" % (float(np.mean(iters)), worst_T))
    # VERDICT round 5 (weak 3): "same input => same output".  A pair's transform must not depend on the list it is in: the same pairs alone, in a short
    # list (other points per workgroup: icp_pts), in reverse order, in small groups (ER_ICP_GROUP) and through er_registration_batch's shares -- every
    # float32 of every transform and every iteration count identical (round 6: the sums across waves are 64-bit fixed point, k_icp_iter).
    from elasticreconstruction_amd.icp import icp_align, registration_batch
    sel = [0, 7, 23, 49]
    T32 = [T.astype(np.float32) for _, _, T in pairs]
    for k in sel:
        F1, it1, c1, _ = icp_align(srcs[k], tgts[k], T32[k], 0.03, 20, 1e-6, 0)
        assert it1 == int(iters[k]) and np.array_equal(F1.view(np.uint32), fins[k].view(np.uint32)), "pair %d alone differs from the pair in the 50-pair list" % k
    f3, i3, _, _ = icp_align_batch([srcs[k] for k in sel[::-1]], [tgts[k] for k in sel[::-1]], [T32[k] for k in sel[::-1]], 0.03, 20, 1e-6, 0)
    for q, k in enumerate(sel[::-1]):
        assert int(i3[q]) == int(iters[k]) and np.array_equal(f3[q].view(np.uint32), fins[k].view(np.uint32)), "pair %d in a 4-pair list differs" % k
    os.environ["ER_ICP_GROUP"] = "7"
    try:
        f7, i7, _, _ = icp_align_batch(srcs, tgts, T32, 0.03, 20, 1e-6, 0)
    finally:
        del os.environ["ER_ICP_GROUP"]
    assert np.array_equal(np.asarray(i7), np.asarray(iters)) and np.array_equal(np.asarray(f7).view(np.uint32), np.asarray(fins).view(np.uint32))
    fused = registration_batch(srcs, tgts, [T for _, _, T in pairs], 0.03, 40000, 0.25, 20, 1e-6, 0, 0.015, 0.8660, want_info=False)
    assert np.array_equal(np.asarray(fused["T"], np.float32).view(np.uint32), np.asarray(fins).view(np.uint32)), "er_registration_batch's shares change a transform"


def test_hard_pairs_at_config2_size_equal_the_reference_ccorresapp(gpu, tmp_path):
    """the HARD list bench.py times (icp.hard_set) -- the configs[2] fragments (250 k points) with guesses up to
    6 deg / 6 cm off the ground truth, so that the 20-iteration limit, the transform criterion and pairs that walk AWAY from the truth
    are on the path (BuildCorrespondence/CorresApp.cpp:295-312) -- through the batch entry points, then >= 8 of its pairs (every pair
    at the iteration limit, the one that ends farthest from the ground truth, the slowest converging one, then the first ones) against
    the reference's own compiled CCorresApp (oracle/_ref/libref_corres.so; tests/corres_helpers.py::check_pairs_against_reference):
    pre-check counts and the accept rule, iteration counts and converged flags EXACT, |dT| <= 1e-5 against both CCorresApp::Registration
    and the reference-side ICP, corres_<i>_<j>.txt byte for byte from the HIP transforms, information matrices to 1e-9.  Where the
    reference build did not travel, the same quantities are compared with the restatement (oracle/icp_oracle.cpp)."""
    from corres_helpers import check_pairs_against_reference, hard_pair_list, select_hard
    from elasticreconstruction_amd.icp import count_inliers_batch, find_correspondence_batch, icp_align_batch
    from oracle.pyoracle import RefCorres
    n_frag, n_pairs = 25, 50
    frs = synth.fragment_set(n_frag, 250000, device="cuda:0")
    gc = [Cloud(x, n, 0.03) for x, n, _ in frs]
    pairs = hard_pair_list(frs, n_pairs)
    srcs, tgts = [gc[b] for _, b, _ in pairs], [gc[a] for a, _, _ in pairs]
    cnts = count_inliers_batch(srcs, tgts, [T for _, _, T in pairs], 0.03)
    fins, iters, conv, _ = icp_align_batch(srcs, tgts, [T.astype(np.float32) for _, _, T in pairs], 0.03, 20, 1e-6, 0)
    lists, infos = find_correspondence_batch(srcs, tgts, [F.astype(np.float64) for F in fins], 0.015, 0.8660, True)
    gt_err = [float(np.abs(F.astype(np.float64) - np.linalg.inv(frs[a][2]) @ frs[b][2]).max()) for F, (a, b, _) in zip(fins, pairs)]
    sel = select_hard(iters, gt_err, want=8)
    assert len(sel) >= 8 and max(int(i) for i in iters) >= 12, "the hard list is not hard: %s" % [int(i) for i in iters]
    if RefCorres.available():
        out = check_pairs_against_reference(frs, pairs, sel, cnts, fins, iters, conv, lists, infos, str(tmp_path))
        assert out["pairs"] >= 8
    else:
        oc = {q: IcpOracle(frs[q][0], frs[q][1], 0.03) for k in sel for q in pairs[k][:2]}
        out = {"against": "oracle/icp_oracle.cpp (the reference build did not travel)", "selected": sel}
        for k in sel:
            a, b, T = pairs[k]
            assert int(cnts[k]) == oc[b].count_inliers(oc[a], T, 0.03)
            To, ito, co, _ = oc[b].align(oc[a], T.astype(np.float32), 0.03, 20, 1e-6, 0)
            assert (int(iters[k]), bool(conv[k])) == (ito, co) and np.abs(fins[k] - To).max() <= TOL_T, "pair %d" % k
            po, io = oc[b].find_correspondence(oc[a], fins[k].astype(np.float64), 0.015, 0.8660, want_info=True)
            assert np.array_equal(lists[k], po) and np.allclose(infos[k], io, rtol=1e-9, atol=1e-6)
    print("hard list: iterations %s, converged %d / %d, max ground-truth error %.3g (pair %d); checked: %s"
          % ([int(i) for i in iters], int(np.sum(conv)), n_pairs, max(gt_err), int(np.argmax(gt_err)), out))
    for c in gc:
        c.close()


def one_iteration_at_a_time(src, tgt, o_src, o_tgt, guess, what, states=(0, 1, 5, 12, 19), tol=1e-6):
    """A loop that uses up PCL's 20 iterations, compared STEP BY STEP: from the oracle's own state after j iterations both sides run ONE iteration and must
    land within 1e-6 of each other -- the loop body is pinned where the whole loop, still moving when it is cut off, only allows 1e-4."""
    for j in states:
        G = guess if j == 0 else o_src.align(o_tgt, guess, 0.03, j, 1e-6, 0)[0]
        To, ito, _, _ = o_src.align(o_tgt, np.asarray(G, np.float32), 0.03, 1, 1e-6, 0)
        Tg, itg, _, _ = icp_align(src, tgt, np.asarray(G, np.float32), 0.03, 1, 1e-6, 0)
        assert itg == ito == 1, "%s, one iteration from state %d: %d vs %d iterations" % (what, j, itg, ito)
        assert np.abs(Tg - To).max() <= tol, "%s, one iteration from state %d: |dT| = %.3g" % (what, j, np.abs(Tg - To).max())


def test_kinfu_like_fragments_at_config2_size(gpu, tmp_path):
    """the configs[2]-size list on fragments that look like cloud_bin_<i>.pcd -- synth.kinfu_fragment: 50 depth frames of a
    hand-held sweep integrated into a TSDF volume by this library, zero crossings extracted, normals = the normalised TSDF gradient (NaN at
    the border of the observed region, filtered like CCorresApp::LoadData does, CorresApp.cpp:93-97), thinned ~ 1 / z^2 (> 10 : 1 density
    contrast inside one cloud, dozens of points in a near cell, one or two in a far one), odd fragments from depth images with 2 mm noise.
    (a) 50 pairs over 25 fragments, guesses <= 2 deg / 2 cm off: every pair against the CPU oracle -- inlier counts, iteration counts,
        converged flags and correspondence index lists EXACT, transforms within 1e-5, information within 1e-9;
    (b) the same fragments with guesses up to 4 deg / 4 cm off (half of the pairs are then rejected by the pre-check, the others need up to ~14
        iterations): the selected pairs (iteration limit, farthest from the truth, ... plus six accepted ones) against the reference's own compiled
        CCorresApp (oracle/_ref/libref_corres.so): pre-check count and accept rule for all of them, the ICP loop, the correspondence file byte for byte
        and the information matrix for the accepted ones."""
    from corres_helpers import check_pairs_against_reference, hard_pair_list, select_hard
    from elasticreconstruction_amd.icp import count_inliers_batch, find_correspondence_batch, icp_align_batch
    from oracle.pyoracle import RefCorres
    n_frag, n_pairs = 25, 50
    frs, stats = [], []
    for i in range(n_frag):                                   # sweeps 7.2 degrees apart (25 of 50 around the room): a pair's fragments are 7 - 22 degrees apart
        x, n, F, st = synth.kinfu_fragment(i, 2 * n_frag, 250000, noise_mm=2.0 if i % 2 else 0.0)
        ok = ~np.isnan(n).any(axis=1)
        frs.append((np.ascontiguousarray(x[ok]), np.ascontiguousarray(n[ok]), F))
        stats.append(st)
    nan_frac = float(np.mean([st["nan_fraction"] for st in stats]))
    occ = [synth.cell_occupancy(x) for x, _, _ in frs]
    assert all(len(x) > 150000 for x, _, _ in frs), [len(x) for x, _, _ in frs]
    assert 0.005 < nan_frac < 0.3, "NaN normals: %.3f of the points" % nan_frac
    assert max(o[0] for o in occ) >= 10 * np.mean([o[1] for o in occ]) or max(o[0] for o in occ) >= 40, occ[:3]      # the density contrast is there
    gc = [Cloud(x, n, 0.03) for x, n, _ in frs]
    oc = [IcpOracle(x, n, 0.03) for x, n, _ in frs]
    pairs = synth.chain_pair_list(frs, n_pairs, 2.0, 0.02, 700)          # neighbours 1 (24 pairs), 2 (23) and 3 (3) sweeps apart: the path is open, no wrap-around
    assert len(pairs) == n_pairs and len({q for a, b, _ in pairs for q in (a, b)}) == n_frag
    srcs, tgts = [gc[b] for _, b, _ in pairs], [gc[a] for a, _, _ in pairs]
    cnts = count_inliers_batch(srcs, tgts, [T for _, _, T in pairs], 0.03)
    fins, iters, conv, _ = icp_align_batch(srcs, tgts, [T.astype(np.float32) for _, _, T in pairs], 0.03, 20, 1e-6, 0)
    lists, infos = find_correspondence_batch(srcs, tgts, [F.astype(np.float64) for F in fins], 0.015, 0.8660, True)
    worst_T, gt = 0.0, []
    for k, (a, b, T) in enumerate(pairs):
        assert int(cnts[k]) == oc[b].count_inliers(oc[a], T, 0.03), "pair %d: inlier count" % k
        To, ito, co, _ = oc[b].align(oc[a], T.astype(np.float32), 0.03, 20, 1e-6, 0)
        assert (int(iters[k]), bool(conv[k])) == (ito, co), "pair %d: iterations/converged %s vs %s" % (k, (iters[k], conv[k]), (ito, co))
        worst_T = max(worst_T, float(np.abs(fins[k] - To).max()))
        # (a pair that uses up the 20 iterations was stopped while still moving: no fixed point contracts the one-ulp differences of the float32
        #  increments -- first seen on fragments 29 degrees apart: 2e-5 after 20 iterations.  Round 6: 1e-4 end to end -- five times the
        #  observed difference, not the 1e-3 of round 5 -- AND every sampled iteration of such a loop by itself, from the oracle's own state, within 1e-6)
        assert np.abs(fins[k] - To).max() <= (1e-4 if ito >= 20 else TOL_T), "pair %d (%d iterations): transform differs by %.3g" % (k, ito, np.abs(fins[k] - To).max())
        if ito >= 20:
            one_iteration_at_a_time(srcs[k], tgts[k], oc[b], oc[a], T.astype(np.float32), "pair %d" % k)
        po, io = oc[b].find_correspondence(oc[a], fins[k].astype(np.float64), 0.015, 0.8660, want_info=True)
        assert np.array_equal(lists[k], po), "pair %d: %d vs %d correspondences" % (k, lists[k].shape[0], po.shape[0])
        assert np.allclose(infos[k], io, rtol=1e-9, atol=1e-6)
        gt.append(float(np.abs(fins[k].astype(np.float64) - np.linalg.inv(frs[a][2]) @ frs[b][2]).max()))
    # extracted surfaces of two different sweeps: the ICP fixed point sits within a voxel of the ground truth (5.9 mm), not on it
    # (noisy estimated normals make a few pairs crawl: such a pair may use up PCL's 20 iterations even from a 2 deg / 2 cm guess -- in the reference exactly as here)
    assert int(np.min(iters)) >= 1 and int(np.sum(np.asarray(iters) >= 20)) <= 3, "iterations: %s" % [int(i) for i in iters]
    assert np.median(gt) < 3e-3 and np.sort(gt)[-4] < 2e-2, "ground truth missed: median %.3g, fourth largest %.3g" % (np.median(gt), np.sort(gt)[-4])
    print("kinfu-like list: %d pairs, %.0f points per fragment after the NaN filter (%.1f %% NaN normals), cells max / mean occupancy %d / %.1f, "
          "mean %.2f ICP iterations (max %d), max |T_gpu - T_oracle| = %.2g, ground-truth error median %.2g max %.2g"
          % (n_pairs, np.mean([len(x) for x, _, _ in frs]), 100 * nan_frac, max(o[0] for o in occ), np.mean([o[1] for o in occ]),
             float(np.mean(iters)), int(np.max(iters)), worst_T, np.median(gt), max(gt)))
    # (b) the hard guesses on the same fragments against the reference's own code
    hard = synth.chain_pair_list(frs, n_pairs, 4.0, 0.04, 1700)         # (at 6 deg / 6 cm the pre-check rejects five of six pairs of these fragments: nothing for Registration to run)
    h_cnts = count_inliers_batch(srcs, tgts, [T for _, _, T in hard], 0.03)
    h_fins, h_iters, h_conv, _ = icp_align_batch(srcs, tgts, [T.astype(np.float32) for _, _, T in hard], 0.03, 20, 1e-6, 0)
    h_lists, h_infos = find_correspondence_batch(srcs, tgts, [F.astype(np.float64) for F in h_fins], 0.015, 0.8660, True)
    h_err = [float(np.abs(F.astype(np.float64) - np.linalg.inv(frs[a][2]) @ frs[b][2]).max()) for F, (a, b, _) in zip(h_fins, hard)]
    sel = select_hard(h_iters, h_err, want=8)
    if RefCorres.available():
        # ICP loops are compared for the pairs the pre-check ACCEPTS (what Registration runs); a rejected pair -- pair (3, 4) of this list: 4410 of 242 k points
        # with a neighbour -- is compared up to its rejection: its loop is chaotic (profiles/r05f_icp_trace_rejected_pair.txt)
        acc = [k for k in range(n_pairs) if int(h_cnts[k]) >= 40000 or min(h_cnts[k] / float(len(frs[hard[k][0]][0])), h_cnts[k] / float(len(frs[hard[k][1]][0]))) > 0.25]
        sel = sorted(set(sel) | set(acc[:6]))
        # (round 6: pairs at the iteration limit 1e-4 end to end and step by step at 1e-6; the rejected pairs' loops, chaotic as wholes, step by step too)
        step = lambda k, G: icp_align(srcs[k], tgts[k], G, 0.03, 1, 1e-6, 0)[:3]
        out = check_pairs_against_reference(frs, hard, sel, h_cnts, h_fins, h_iters, h_conv, h_lists, h_infos, str(tmp_path), tol_T_at_limit=1e-4, icp_on_rejected=False,
                                            step_fn=step)
        assert out["pairs"] >= 8 and out["icp_loops_compared"] >= 4 and out["single_iterations_compared"] >= 3, out
    else:
        out = {"against": "oracle/icp_oracle.cpp (the reference build did not travel)", "selected": sel}
        for k in sel:
            a, b, T = hard[k]
            assert int(h_cnts[k]) == oc[b].count_inliers(oc[a], T, 0.03)
            To, ito, co, _ = oc[b].align(oc[a], T.astype(np.float32), 0.03, 20, 1e-6, 0)
            assert (int(h_iters[k]), bool(h_conv[k])) == (ito, co) and np.abs(h_fins[k] - To).max() <= (1e-4 if ito >= 20 else TOL_T), "hard pair %d" % k
            if ito >= 20:
                one_iteration_at_a_time(srcs[k], tgts[k], oc[b], oc[a], T.astype(np.float32), "hard pair %d" % k)
            po, io = oc[b].find_correspondence(oc[a], h_fins[k].astype(np.float64), 0.015, 0.8660, want_info=True)
            assert np.array_equal(h_lists[k], po) and np.allclose(h_infos[k], io, rtol=1e-9, atol=1e-6)
    print("kinfu-like hard list: iterations %s, converged %d / %d; checked: %s" % ([int(i) for i in h_iters], int(np.sum(h_conv)), n_pairs, out))


def test_registration_batch_equals_the_three_stage_calls(gpu, monkeypatch):
    """er_registration_batch (Registration + FindCorrespondence of a pair list in one call, the shares on host threads of their own) against
    the three *_batch calls in sequence with the accept rule of CorresApp.cpp:270 applied in between: pre-check counts, accept flags,
    iteration counts, converged flags and correspondence lists EXACT, transforms within 1e-6 (a share is a smaller group: k_icp_iter may
    take fewer points per thread, which reorders its float64 sums), information matrices to 1e-9 -- with one, two and three shares, a
    hopeless pair (rejected by the pre-check: no ICP, no list, its transform stays the guess), an empty source and shared clouds."""
    from elasticreconstruction_amd.icp import count_inliers_batch, find_correspondence_batch, icp_align_batch, registration_batch
    data = [make_pair(n=60000 + 9000 * i, rot=1.0 + 0.3 * i, trans=0.01, seed=40 + i) for i in range(3)]
    cl = [(Cloud(x0, n0, 0.03), Cloud(x1, n1, 0.03), P) for (x0, n0), (x1, n1), P in data]
    empty = Cloud(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), 0.03)
    srcs, tgts, Ts = [], [], []
    for k in range(19):
        tgt, src, P = cl[k % 3]
        srcs.append(src); tgts.append(tgt)
        Ts.append(P @ synth.perturbation(80 + k, 0.8 + 0.2 * (k % 5), 0.006) if k not in (4, 13) else synth.perturbation(3 + k, 40, 1.0))
    srcs.append(empty); tgts.append(cl[0][0]); Ts.append(np.eye(4))
    n = len(srcs)
    reg_num, reg_ratio = 20000, 0.25
    cnts = count_inliers_batch(srcs, tgts, Ts, 0.03)
    ns, nt = np.array([max(len(s), 1) for s in srcs], float), np.array([len(t) for t in tgts], float)
    acc = (cnts >= reg_num) | ((cnts / nt > reg_ratio) & (cnts / ns > reg_ratio))
    acc &= np.array([len(s) > 0 for s in srcs])
    ai = np.nonzero(acc)[0]
    assert not acc[4] and not acc[13] and not acc[-1] and acc.sum() == n - 3
    F, its, cv, _ = icp_align_batch([srcs[k] for k in ai], [tgts[k] for k in ai], [Ts[k].astype(np.float32) for k in ai], 0.03, 20, 1e-6, 0)
    lists, infos = find_correspondence_batch([srcs[k] for k in ai], [tgts[k] for k in ai], [f.astype(np.float64) for f in F], 0.015, 0.8660, True)
    for shares in ("1", "2", "3"):
        monkeypatch.setenv("ER_ICP_SHARES", shares)
        r = registration_batch(srcs, tgts, Ts, 0.03, reg_num, reg_ratio, 20, 1e-6, 0, 0.015, 0.8660, want_info=True)
        assert np.array_equal(r["counts"], cnts) and np.array_equal(r["accepted"], acc), shares
        for q, k in enumerate(ai):
            assert (int(r["iterations"][k]), bool(r["converged"][k])) == (int(its[q]), bool(cv[q])), (shares, k)
            assert np.abs(r["T"][k] - F[q]).max() <= 1e-6, (shares, k, np.abs(r["T"][k] - F[q]).max())
            # the list is taken at the call's OWN final transform: identical whenever that transform is (it is, bit for bit, in practice)
            if np.array_equal(r["T"][k], F[q]):
                assert np.array_equal(r["lists"][k], lists[q]), (shares, k)
                assert np.allclose(r["info"][k], infos[q], rtol=1e-9, atol=1e-6)
            else:
                assert abs(len(r["lists"][k]) - len(lists[q])) <= max(3, len(lists[q]) // 1000)
        for k in np.nonzero(~acc)[0]:
            assert r["iterations"][k] == 0 and not r["converged"][k] and len(r["lists"][k]) == 0 and not r["info"][k].any()
            assert np.array_equal(r["T"][k], Ts[k].astype(np.float32))
    assert sum(len(l) for l in lists) > 100000


def test_cloud_create_batch_equals_single_creates(gpu):
    """er_cloud_create_batch (chunks of up to 8 clouds that share their allocations and ONE set of grid launches; all uploads queued up front) builds
    the SAME clouds as er_cloud_create one by one: 11 fragments of different sizes (two chunks), one of them empty, the input arrays once in
    pageable and once in page-locked memory; every pair search through them gives the single-create results exactly.  A non-finite
    coordinate anywhere in the list fails the whole call and leaves no cloud behind."""
    from elasticreconstruction_amd import _ffi
    from elasticreconstruction_amd.icp import count_inliers_batch, find_correspondence_batch
    frs = synth.fragment_set(5, 60000, seed=9)
    arrays = [(x[:60000 - 4000 * k], n[:60000 - 4000 * k]) for k, (x, n, _) in enumerate(frs)] + [(frs[0][0][:777], frs[0][1][:777])]
    arrays += [(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))] + [(frs[k][0][5:30005], frs[k][1][5:30005]) for k in range(4)]
    assert len(arrays) == 11
    single = [Cloud(x, n, 0.03) for x, n in arrays]
    arena = _ffi.PinnedArena()
    arena.reset(sum(x.nbytes + n.nbytes for x, n in arrays) + 8192 * 2 * len(arrays))
    pinned = []
    for x, n in arrays:
        px, pn = arena.take(x.shape, np.float32), arena.take(n.shape, np.float32)
        px[...] = x
        pn[...] = n
        pinned.append((px, pn))
    T01 = np.linalg.inv(frs[0][2]) @ frs[1][2] @ synth.perturbation(5, 1.0, 0.01)
    for src_arrays in (arrays, pinned):
        batch = Cloud.create_batch(src_arrays, 0.03)
        assert [len(c) for c in batch] == [len(c) for c in single]
        pairs = [(1, 0, T01), (8, 7, np.linalg.inv(frs[0][2]) @ frs[1][2]), (5, 0, np.eye(4)), (6, 0, np.eye(4)), (0, 6, np.eye(4)), (2, 2, np.eye(4))]
        for clouds in (batch,):
            cb = count_inliers_batch([clouds[s] for s, _, _ in pairs], [clouds[t] for _, t, _ in pairs], [T for _, _, T in pairs], 0.03)
            cs = count_inliers_batch([single[s] for s, _, _ in pairs], [single[t] for _, t, _ in pairs], [T for _, _, T in pairs], 0.03)
            assert np.array_equal(cb, cs) and cb[0] > 10000 and cb[3] == 0 and cb[4] == 0 and cb[5] == len(single[2])
            lb, ib = find_correspondence_batch([clouds[s] for s, _, _ in pairs], [clouds[t] for _, t, _ in pairs], [T for _, _, T in pairs], 0.015, 0.8660, True)
            ls, is_ = find_correspondence_batch([single[s] for s, _, _ in pairs], [single[t] for _, t, _ in pairs], [T for _, _, T in pairs], 0.015, 0.8660, True)
            assert all(np.array_equal(a, b) for a, b in zip(lb, ls)) and np.allclose(ib, is_, rtol=1e-9, atol=1e-6)      # (float64 atomics: order-dependent last bits)
        for c in batch:
            c.close()
    # 19 small clouds = three chunks (the first and the third share a compute lane and its scratch), with an empty one, a single point and a
    # cloud whose points all share one cell among them: the same searches through single creates
    rng = np.random.default_rng(77)
    small = []
    for k in range(19):
        m = [0, 1, 300][k] if k < 3 else int(rng.integers(50, 4000))
        sel = rng.choice(len(frs[k % 5][0]), m, replace=False)
        small.append((np.ascontiguousarray(frs[k % 5][0][sel]), np.ascontiguousarray(frs[k % 5][1][sel])))
    small[2] = (np.ascontiguousarray(small[2][0] * np.float32(1e-3) + frs[0][0][0]), small[2][1])      # 300 points within a few millimetres
    s_single = [Cloud(x, n, 0.03) for x, n in small]
    s_batch = Cloud.create_batch(small, 0.03)
    assert [len(c) for c in s_batch] == [len(x) for x, _ in small]
    sp = [(k, (k + 5) % 19, np.eye(4)) for k in range(19)] + [(2, 2, np.eye(4)), (1, 1, np.eye(4)), (0, 4, np.eye(4)), (4, 0, np.eye(4))]
    cb = count_inliers_batch([s_batch[s] for s, _, _ in sp], [s_batch[t] for _, t, _ in sp], [T for _, _, T in sp], 0.03)
    cs = count_inliers_batch([s_single[s] for s, _, _ in sp], [s_single[t] for _, t, _ in sp], [T for _, _, T in sp], 0.03)
    assert np.array_equal(cb, cs) and cb[19] == 300 and cb[20] == 1 and cb[21] == 0 and cb[22] == 0 and cb.sum() > 600
    lb, _ = find_correspondence_batch([s_batch[s] for s, _, _ in sp], [s_batch[t] for _, t, _ in sp], [T for _, _, T in sp], 0.015, 0.0, False)
    ls, _ = find_correspondence_batch([s_single[s] for s, _, _ in sp], [s_single[t] for _, t, _ in sp], [T for _, _, T in sp], 0.015, 0.0, False)
    assert all(np.array_equal(a, b) for a, b in zip(lb, ls)) and len(lb[19]) == 300
    for c in s_batch[::2]:                                        # clouds of a chunk share their allocations: any order of closing is fine
        c.close()
    cb2 = count_inliers_batch([s_batch[1], s_batch[3]], [s_batch[1], s_batch[5]], [np.eye(4)] * 2, 0.03)
    cs2 = count_inliers_batch([s_single[1], s_single[3]], [s_single[1], s_single[5]], [np.eye(4)] * 2, 0.03)
    assert np.array_equal(cb2, cs2) and cb2[0] == 1
    for c in s_batch[1::2] + s_single:
        c.close()
    bad = [(x.copy(), n) for x, n in arrays]
    bad[9][0][123, 1] = np.inf
    with pytest.raises(_ffi.ErError, match="non-finite"):
        Cloud.create_batch(bad, 0.03)
    arena.close()
    for c in single:
        c.close()


def test_nearest_neighbour_straight_behind_a_cell_face(gpu):
    """The pruning margin of nn_block (grid_slack): 40 queries whose nearest neighbour lies just behind a cell face, straight along the axis,
    with a competitor in the query's own cell that is farther by a fraction of a micrometre (tests/nn_margin_cases.py; tests/test_nn_margin.py
    shows in float32 arithmetic that the margin of rounds 1-4 skipped the neighbour's cell for every one of them).  The HIP search names the
    point behind the face, like the oracle's unpruned 27-cell scan."""
    from nn_margin_cases import build
    tgt, src, expect, _, _ = build()
    nt = np.tile(np.array([[0, 0, 1]], np.float32), (len(tgt), 1))
    ns = np.tile(np.array([[0, 0, 1]], np.float32), (len(src), 1))
    ct, cs = Cloud(tgt, nt, 0.03), Cloud(src, ns, 0.03)
    pg, _ = find_correspondence(cs, ct, np.eye(4), 0.015, 0.8660)
    po, _ = IcpOracle(src, ns, 0.03).find_correspondence(IcpOracle(tgt, nt, 0.03), np.eye(4), 0.015, 0.8660)
    assert pg.shape == (40, 2) and np.array_equal(pg, po), "%d of 40 rows differ from the oracle" % int((pg != po).any(1).sum())
    assert np.array_equal(pg[:, 0], expect)
    assert count_inliers(cs, ct, np.eye(4), 0.03) == 40


def _fuzz_cloud(rng, kind, cell):
    """Adversarial inputs for the exact search: what nn_block's pruning, tie rule and cell arithmetic depend on."""
    if kind == 0:                       # lattice with dyadic spacing, points ON cell faces (cell = 2^-5): distance ties, face cases
        m = int(rng.integers(6, 14))
        g = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
        x = g * np.float32(cell / float(rng.choice([1, 2, 4]))) + rng.integers(0, 4, 3).astype(np.float32) * np.float32(cell)
        x = x[rng.random(len(x)) < 0.7]
    elif kind == 1:                     # a few very dense clusters (hundreds of points per cell) + sparse background
        c = rng.uniform(0.2, 1.0, (int(rng.integers(2, 6)), 3))
        x = np.concatenate([c[k] + rng.normal(0, 0.004, (int(rng.integers(200, 1500)), 3)) for k in range(len(c))] + [rng.uniform(0.0, 1.2, (800, 3))]).astype(np.float32)
    elif kind == 2:                     # a noisy sheet far from the origin (float32 resolution 4e-6 m at 40 m), negative coordinates
        uv = rng.uniform(-0.6, 0.6, (int(rng.integers(2000, 9000)), 2))
        x = np.stack([uv[:, 0], uv[:, 1], 0.1 * np.sin(3 * uv[:, 0]) + rng.normal(0, 0.002, len(uv))], 1) + np.array([-40.0, 17.0, 33.0])
        x = x.astype(np.float32)
    else:                               # tiny clouds: 1 .. 5 points, duplicates
        x = rng.uniform(0.0, 0.05, (int(rng.integers(1, 6)), 3)).astype(np.float32)
        x = np.concatenate([x, x[:1]])
    n = rng.normal(0, 1, x.shape)
    n = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(np.float32)
    return np.ascontiguousarray(x), np.ascontiguousarray(n)


def test_randomised_clouds_against_the_oracle(gpu):
    """Seeded fuzz of the exact search against the oracle's unpruned 27-cell scan: lattices whose points lie on cell faces at dyadic spacings (ties ->
    lower index), clusters of hundreds of points per cell, sheets 40 m from the origin, clouds of one to five points; source = an independent cloud of
    the same kind or the target moved by a small rigid motion; radii from a tenth of the cell up to the cell.  Pre-check counts and correspondence
    lists (index pairs in file order) must be EQUAL.  ER_FUZZ_SEED / ER_FUZZ_CASES widen the sweep (one-off runs: profiles/r06x_fuzz_sweep.txt)."""
    rng = np.random.default_rng(int(os.environ.get("ER_FUZZ_SEED", "606")))
    cell = 0.03125
    rows_compared = 0
    for case in range(int(os.environ.get("ER_FUZZ_CASES", "12"))):
        kind = case % 4
        xt, nt = _fuzz_cloud(rng, kind, cell)
        if case % 3 == 0:
            xs, ns = _fuzz_cloud(rng, kind, cell)
        else:
            sel = rng.random(len(xt)) < 0.8
            xs, ns = np.ascontiguousarray(xt[sel]), np.ascontiguousarray(nt[sel])
        c = xt.mean(0).astype(np.float64)
        P = synth.perturbation(int(rng.integers(1 << 30)), float(rng.choice([0.0, 0.5, 5.0])), float(rng.choice([0.0, 0.002, 0.02])))
        Tc = np.eye(4)
        Tc[:3, 3] = c
        T = Tc @ P @ np.linalg.inv(Tc)                                   # the motion about the cloud's centre (a 5 degree turn about the origin moves a sheet at 40 m by metres)
        if case % 5 == 4:
            T = np.eye(4)                                                # exact coincidences: zero distances, duplicates
        tgt, src, otgt, osrc = Cloud(xt, nt, cell), Cloud(xs, ns, cell), IcpOracle(xt, nt, cell), IcpOracle(xs, ns, cell)
        for r in (cell, 0.5 * cell, 0.1 * cell):
            cg, co = count_inliers(src, tgt, T, r), osrc.count_inliers(otgt, T, r)
            assert cg == co, "fuzz case %d (kind %d, %d -> %d points, radius %g): count %d vs %d" % (case, kind, len(xs), len(xt), r, cg, co)
            pg, _ = find_correspondence(src, tgt, T, r, -1.0)            # every pair within the radius (no normal filter): the NN index itself
            po, _ = osrc.find_correspondence(otgt, T, r, -1.0)
            assert pg.shape == po.shape and np.array_equal(pg, po), "fuzz case %d (kind %d, radius %g): %d of %d rows differ" % (
                case, kind, r, int((pg != po).any(1).sum()) if pg.shape == po.shape else -1, len(po))
            rows_compared += len(po)
        for h in (tgt, src):
            h.close()
        for h in (otgt, osrc):
            h.close()
    assert rows_compared > 2000, rows_compared                        # (the sweep is not vacuous)
