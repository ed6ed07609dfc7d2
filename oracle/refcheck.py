"""TEST INFRASTRUCTURE (like everything under oracle/): the checker that compares path-B results -- the HIP path's in tests/test_icp_gpu.py
and in bench.py's parity legs, the restatement's in tests/test_corres_reference.py -- with the reference's own compiled code
(oracle/_ref/libref_corres.so = /root/reference/BuildCorrespondence/CorresApp.cpp built in place against oracle/stub_corres).
Nothing under elasticreconstruction_amd/ imports this."""
import os

import numpy as np


def select_hard(iters, gt_err, max_iter=20, want=8):
    """Which pairs of a list go to the (slow) reference: every pair that ran into the iteration limit (at most three), the one
    that ended farthest from the ground truth, the one with the most iterations below the limit, then the first ones."""
    iters = [int(i) for i in iters]
    sel = [k for k in range(len(iters)) if iters[k] >= max_iter][:3]
    sel.append(int(np.argmax(gt_err)))
    below = [k for k in range(len(iters)) if iters[k] < max_iter]
    if below:
        sel.append(max(below, key=lambda k: iters[k]))
    for k in range(len(iters)):
        if len(set(sel)) >= want:
            break
        sel.append(k)
    return sorted(set(sel))


def point_to_plane_conditioning(src_xyz, tgt_nrm, T, corr):
    """lambda_min / lambda_max of the (diagonally scaled) 6 x 6 normal matrix of TransformationEstimationPointToPlaneLLS over the
    correspondences `corr` (rows: target index, source index) at transform T -- how well the pair constrains the six degrees of
    freedom.  A fragment pair that shares one wall plus floor and ceiling slides along the wall: the matrix is singular to
    working precision and the solve returns rounding noise along that direction, in the reference exactly as here."""
    c = np.asarray(corr)
    if len(c) < 6:
        return 0.0
    s = np.asarray(src_xyz, np.float64)[c[:, 1]] @ np.asarray(T, np.float64)[:3, :3].T + np.asarray(T, np.float64)[:3, 3]
    n = np.asarray(tgt_nrm, np.float64)[c[:, 0]]
    A = np.concatenate([np.cross(s, n), n], axis=1)                    # rows (s x n, n): PCL's a, b, c, nx, ny, nz
    M = A.T @ A
    d = np.sqrt(np.clip(np.diag(M), 1e-300, None))
    w = np.linalg.eigvalsh(M / d[:, None] / d[None, :])
    return float(max(w[0], 0.0) / w[-1])


def check_pairs_against_reference(frs, pairs, sel, cnts, fins, iters, conv, lists, infos, tmp_dir, reg_dist=0.03, tol_T=1e-5, reg_num=40000,
                                  reg_ratio=0.25):
    """The results somebody (the HIP path; in the CPU suite: the restatement) produced for pairs[k], k in sel -- pre-check count,
    final transform, iteration count, converged flag, correspondence list and information matrix AT that final transform -- against
    the reference's own compiled code: CCorresApp::Registration (pre-check count = frame_ and the accept rule, CorresApp.cpp:257-281;
    final transform of the accepted pairs), the stub's PCL 1.7 ICP called as CorresApp.cpp:295-306 configures it on EVERY selected
    pair (iteration count, converged, transform -- a 6 cm guess can fall below the pre-check, the ICP loop is still compared), and
    CCorresApp::FindCorrespondence run from the candidate's final transforms of the accepted pairs (corres_<i>_<j>.txt byte for byte,
    frame_, information).  Returns a summary dict."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.pyoracle import RefCorres
    need = sorted({q for k in sel for q in pairs[k][:2]})
    app = RefCorres(reg_dist=reg_dist, reg_num=reg_num, reg_ratio=reg_ratio)
    idx = {q: app.add_cloud(frs[q][0], frs[q][1]) for q in need}
    for k in sel:
        a, b, T = pairs[k]
        app.add_pair(idx[a], idx[b], len(need), T)
    app.Registration()
    reg = app.pairs()
    app.close()
    worst_T, accepted, degenerate = 0.0, [], {}

    def ill_posed(k):
        """True (and recorded) if pair k's point-to-plane system is singular to working precision at the candidate's transform."""
        if k not in degenerate:
            a, b, _ = pairs[k]
            c = point_to_plane_conditioning(frs[b][0], frs[a][1], fins[k], lists[k]) if lists[k] is not None else 1.0
            if c < 1e-7:
                degenerate[k] = c
        return k in degenerate
    for k, (ri, rj, rframe, rT, _) in zip(sel, reg):
        a, b, T = pairs[k]
        c = int(cnts[k])
        acc = c >= reg_num or (c / float(len(frs[a][0])) > reg_ratio and c / float(len(frs[b][0])) > reg_ratio)
        assert rframe == (c if acc else -1), "pair %d: pre-check count %d (accept %s), reference frame_ %d" % (k, c, acc, rframe)
        if not acc:
            continue
        accepted.append((k, rframe))
        d = float(np.abs(rT - np.asarray(fins[k], np.float64)).max())
        if d > tol_T and ill_posed(k):
            continue                                                   # no well-defined answer to compare (listed in the summary)
        worst_T = max(worst_T, d)
        assert d <= tol_T, "pair %d (%d iterations): transform differs from CCorresApp's by %.3g" % (k, int(iters[k]), d)

    def ref_icp(k):
        a, b, T = pairs[k]
        return RefCorres.icp(frs[b][0], frs[b][1], frs[a][0], frs[a][1], T.astype(np.float32), reg_dist, 20, 1e-6)
    with_icp = [k for k in sel if fins[k] is not None]                 # (a candidate that follows the reference's flow has no ICP result for rejected pairs)
    assert all(fins[k] is not None for k, _ in accepted), "an accepted pair without a final transform"
    with ThreadPoolExecutor(8) as ex:                                  # (ctypes releases the GIL; the stub's ICP is single-threaded)
        ricp = list(ex.map(ref_icp, with_icp))
    for k, (T1, it1, c1, _) in zip(with_icp, ricp):
        if k in degenerate:
            continue
        assert (int(iters[k]), bool(conv[k])) == (it1, c1), "pair %d: iterations / converged %s, reference %s" % (k, (int(iters[k]), bool(conv[k])), (it1, c1))
        d = float(np.abs(T1.astype(np.float64) - np.asarray(fins[k], np.float64)).max())
        worst_T = max(worst_T, d)
        assert d <= tol_T, "pair %d (%d iterations): transform differs from the reference ICP's by %.3g" % (k, it1, d)
    d = tmp_dir if tmp_dir.endswith("/") else tmp_dir + "/"
    app2 = RefCorres(out_dir=d, reg_dist=reg_dist, reg_num=reg_num, reg_ratio=reg_ratio)
    for q in need:
        assert app2.add_cloud(frs[q][0], frs[q][1]) == idx[q]
    for k, rframe in accepted:
        a, b, _ = pairs[k]
        app2.add_pair(idx[a], idx[b], rframe, np.asarray(fins[k], np.float64))        # frame_ as Registration left it (the ratio test divides by it)
    app2.FindCorrespondence()
    got = app2.pairs()
    app2.close()
    n_rows = 0
    for (k, cnt), (ri, rj, rframe, _, rinfo) in zip(accepted, got):
        text = "".join("%d %d\n" % (p, q) for p, q in np.asarray(lists[k]))
        path = d + "corres_%d_%d.txt" % (ri, rj)
        assert open(path).read() == text, "pair %d: correspondence list differs from the reference's file" % k
        if len(lists[k]) / float(cnt) < 0.5 and reg_num > 0:           # `Reduced too much` (CorresApp.cpp:164-173)
            assert rframe == -1, "pair %d" % k
            continue
        assert rframe == len(lists[k])
        assert np.allclose(rinfo, infos[k], rtol=1e-9, atol=1e-6), "pair %d: information matrix" % k
        n_rows += len(lists[k])
    return {"pairs": len(sel), "selected": [int(k) for k in sel], "accepted_by_the_pre_check": len(accepted),
            "icp_loops_compared": len([k for k in with_icp if k not in degenerate]),
            "ill_posed_pairs_not_compared": {int(k): {"lambda_min_over_lambda_max": v, "why": "the pair's point-to-plane normal matrix is singular to working "
                                                      "precision (one wall + floor + ceiling: the fragments slide along the wall); the solve returns rounding noise "
                                                      "along that direction in the reference as here -- pre-check count and correspondence file at the candidate's "
                                                      "transform are still compared"} for k, v in degenerate.items()},
            "iterations": [int(iters[k]) for k in with_icp], "converged": [bool(conv[k]) for k in with_icp], "max_abs_T_diff": worst_T,
            "tolerance_T": tol_T, "correspondence_rows_compared": n_rows,
            "against": "the reference's CCorresApp compiled in place (oracle/_ref/libref_corres.so): pre-check count + accept rule, ICP iteration "
                       "count / converged / transform, corres_<i>_<j>.txt byte for byte, information matrix"}
