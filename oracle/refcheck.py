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


def point_to_plane_conditioning_at(frs, pair, G, reg_dist):
    """point_to_plane_conditioning of a pair at state G with the correspondences an ICP iteration would use there (exact NN within reg_dist, cKDTree)."""
    from scipy.spatial import cKDTree
    a, b, _ = pair
    G = np.asarray(G, np.float64)
    q = np.asarray(frs[b][0], np.float64) @ G[:3, :3].T + G[:3, 3]
    dist, j = cKDTree(np.asarray(frs[a][0], np.float64)).query(q, distance_upper_bound=reg_dist)
    ok = np.isfinite(dist)
    corr = np.stack([j[ok], np.nonzero(ok)[0]], 1)
    return point_to_plane_conditioning(frs[b][0], frs[a][1], G, corr)


def check_pairs_against_reference(frs, pairs, sel, cnts, fins, iters, conv, lists, infos, tmp_dir, reg_dist=0.03, tol_T=1e-5, reg_num=40000,
                                  reg_ratio=0.25, tol_T_at_limit=None, icp_on_rejected=True, step_fn=None):
    """The results somebody (the HIP path; in the CPU suite: the restatement) produced for pairs[k], k in sel -- pre-check count,
    final transform, iteration count, converged flag, correspondence list and information matrix AT that final transform -- against
    the reference's own compiled code: CCorresApp::Registration (pre-check count = frame_ and the accept rule, CorresApp.cpp:257-281;
    final transform of the accepted pairs), the stub's PCL 1.7 ICP called as CorresApp.cpp:295-306 configures it on EVERY selected
    pair (iteration count, converged, transform -- a 6 cm guess can fall below the pre-check, the ICP loop is still compared), and
    CCorresApp::FindCorrespondence run from the candidate's final transforms of the accepted pairs (corres_<i>_<j>.txt byte for byte,
    frame_, information).  tol_T_at_limit (default: tol_T) applies to pairs that used up PCL's 20 iterations: such a pair was stopped while still
    moving, and a loop that is not at a fixed point carries a one-ulp difference of a float32 increment forward instead of contracting it (noisy
    kinfu-like fragments: 2e-5 after 20 iterations; the uniform fragments stay below 1e-6 even there).  icp_on_rejected=False compares ICP loops only
    for the pairs the pre-check accepts -- what CCorresApp::Registration actually runs: from a guess that leaves 2 % of the points with a neighbour the
    loop solves near-singular systems, the estimate jumps by decimetres per iteration and a 1e-16 difference in the float64 sums grows tenfold per
    iteration (profiles/r05f_icp_trace_rejected_pair.txt).
    step_fn(k, guess float32 4x4) -> (T, iterations, converged): the candidate's ICP limited to ONE iteration (round 6).  When given, the pairs a
    whole-loop comparison is weakest on -- those that use up the 20 iterations, and the rejected ones whose loops are skipped -- are compared STEP BY STEP
    instead: from states of the reference's own trajectory (after 0, 1, 5, 12, 19 iterations; 0, 1, 2 for a rejected pair) both sides run one iteration
    and must land within 1e-6 (1e-5 for a rejected pair) of each other: an error in the loop body cannot hide behind the loop's sensitivity.  Returns a summary dict."""
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
    tol_lim = tol_T if tol_T_at_limit is None else tol_T_at_limit

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
        assert d <= (tol_lim if int(iters[k]) >= 20 else tol_T), "pair %d (%d iterations): transform differs from CCorresApp's by %.3g" % (k, int(iters[k]), d)

    def ref_icp(k):
        a, b, T = pairs[k]
        return RefCorres.icp(frs[b][0], frs[b][1], frs[a][0], frs[a][1], T.astype(np.float32), reg_dist, 20, 1e-6)
    with_icp = [k for k in sel if fins[k] is not None]                 # (a candidate that follows the reference's flow has no ICP result for rejected pairs)
    if not icp_on_rejected:
        with_icp = [k for k in with_icp if k in {q for q, _ in accepted}]
    assert all(fins[k] is not None for k, _ in accepted), "an accepted pair without a final transform"
    with ThreadPoolExecutor(8) as ex:                                  # (ctypes releases the GIL; the stub's ICP is single-threaded)
        ricp = list(ex.map(ref_icp, with_icp))
    for k, (T1, it1, c1, _) in zip(with_icp, ricp):
        if k in degenerate:
            continue
        assert (int(iters[k]), bool(conv[k])) == (it1, c1), "pair %d: iterations / converged %s, reference %s" % (k, (int(iters[k]), bool(conv[k])), (it1, c1))
        d = float(np.abs(T1.astype(np.float64) - np.asarray(fins[k], np.float64)).max())
        worst_T = max(worst_T, d)
        assert d <= (tol_lim if it1 >= 20 else tol_T), "pair %d (%d iterations): transform differs from the reference ICP's by %.3g" % (k, it1, d)
    steps_compared, worst_step = 0, 0.0
    if step_fn is not None:
        acc_set = {q for q, _ in accepted}
        at_limit = [k for k in with_icp if int(iters[k]) >= 20 and k not in degenerate]
        rejected = [k for k in sel if k not in acc_set]
        for k, states, tol in [(k, (0, 1, 5, 12, 19), 1e-6) for k in at_limit] + [(k, (0, 1, 2), 1e-5) for k in rejected]:
            a, b, T = pairs[k]
            for j in states:
                if j == 0:
                    G = T.astype(np.float32)
                else:
                    G, itj, _, _ = RefCorres.icp(frs[b][0], frs[b][1], frs[a][0], frs[a][1], T.astype(np.float32), reg_dist, j, 1e-6)
                    if itj < j:
                        break                                          # the reference's loop ended before state j
                Tr, itr, cr, _ = RefCorres.icp(frs[b][0], frs[b][1], frs[a][0], frs[a][1], np.asarray(G, np.float32), reg_dist, 1, 1e-6)
                Tc, itc, cc = step_fn(k, np.asarray(G, np.float32))
                assert int(itc) == int(itr), "pair %d, one iteration from the reference's state %d: %d iterations, reference %d" % (k, j, itc, itr)
                dstep = float(np.abs(np.asarray(Tc, np.float64) - Tr.astype(np.float64)).max())
                if dstep > tol and point_to_plane_conditioning_at(frs, pairs[k], G, reg_dist) < 1e-7:
                    continue                                           # (a singular system has no answer to compare)
                assert dstep <= tol, "pair %d, one iteration from the reference's state %d: |dT| = %.3g" % (k, j, dstep)
                worst_step = max(worst_step, dstep)
                steps_compared += 1
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
            "tolerance_T": tol_T, "tolerance_T_at_the_iteration_limit": tol_lim, "single_iterations_compared": steps_compared, "max_abs_T_diff_single_iteration": worst_step,
            "correspondence_rows_compared": n_rows,
            "against": "the reference's CCorresApp compiled in place (oracle/_ref/libref_corres.so): pre-check count + accept rule, ICP iteration "
                       "count / converged / transform, corres_<i>_<j>.txt byte for byte, information matrix"}


# ---- path A: a sampled stream of a long Integrate job against the reference's own CIntegrateApp ---------------------------------------
def write_integrate_files(sc, files_dir):
    """pose.log / seg.log / g.ctr of a WHOLE scenario (elasticreconstruction_amd.synth.make_scenario) the way Integrate.exe reads them
    (Integrate/IntegrateApp.cpp:43-79): one pose per fragment, one seg entry per frame, every control lattice; one extra fragment of
    entries at the end so that the last frame still passes `frame_id_ >= traj_.data_.size()` (IntegrateApp.cpp:200-203).
    Returns the three paths."""
    from elasticreconstruction_amd import formats
    interval, n = sc["interval"], sc["n"]
    num = n // interval
    FT = formats.FramedTransformation
    p = [os.path.join(files_dir, f) for f in ("pose.log", "seg.log", "g.ctr")]
    formats.save_log(p[0], [FT(i, i, i + 1, sc["pose"][i]) for i in range(num)] + [FT(num, num, num + 1, sc["pose"][num - 1])])
    formats.save_log(p[1], [FT(i, i, i + 1, sc["seg"][i]) for i in range(n)] +
                     [FT(n + j, n + j, n + j + 1, sc["seg"][n - 1]) for j in range(interval)])
    formats.save_ctr(p[2], sc["grids"][:num])
    return p


def sampled_frames(n_frames, interval, n_runs, run_len):
    """0-based frame indices of n_runs runs of run_len consecutive frames spread evenly over an n_frames job, first run at the head of
    the FIRST fragment, last run at the tail of the LAST one, every run inside one fragment (run_len <= interval)."""
    assert 1 <= run_len <= interval and n_runs >= 2
    num = n_frames // interval
    out = []
    for r in range(n_runs):
        frag = (r * (num - 1)) // (n_runs - 1)
        off = 0 if r == 0 else (interval - run_len if r == n_runs - 1 else ((r * 7) % (interval - run_len + 1)))
        out.extend(range(frag * interval + off, frag * interval + off + run_len))
    return np.array(sorted(set(out)), np.int64)


def reference_volume_of_frames(sc, depth_host, frame_ids0, files_dir, uncapped=False):
    """The volume the REFERENCE's own code (oracle/_ref/libref_tsdf.so = Integrate/*.cpp compiled in place) leaves after the frames
    frame_ids0 (0-based, ascending) of the job `sc`, each through CIntegrateApp::Execute with its TRUE frame id -- so frame f is
    warped with lattice (f / interval) of the job's full .ctr and integrated with traj_[f] of the full trajectory, whatever was
    skipped before it (IntegrateApp.cpp:190-226, 228-268).  depth_host[k] = the uint16 image of frame frame_ids0[k].
    Returns ({unit key: (sdf_, weight_)}, seconds inside Execute, the three file paths)."""
    import time
    from oracle.pyoracle import RefApp
    paths = write_integrate_files(sc, files_dir)
    ref = RefApp(uncapped=uncapped)
    ref.init(pose_traj=paths[0], seg_traj=paths[1], ctr=paths[2], num=sc["n"] // sc["interval"], resolution=sc["resolution"],
             length=sc["length"], interval=sc["interval"])
    t0 = time.perf_counter()
    for k, f in enumerate(frame_ids0):
        ref.execute(int(f) + 1, depth_host[k])
    dt = time.perf_counter() - t0
    vol = {int(k): ref.read_unit(int(k)) for k in ref.unit_keys()}
    ref.close()
    return vol, dt, paths


def unit_coordinates(keys):
    """Unit key -> signed unit lattice coordinates (Integrate/TSDFVolume.h:62-64: key = x * 512 * 512 + y * 512 + z over the index
    shifted by +256 units, TSDFVolume.cpp:50-53) as an int array [n, 3]; coordinates < 0 are units at negative world coordinates."""
    k = np.asarray(keys, np.int64)
    return np.stack([k // (512 * 512) - 256, (k // 512) % 512 - 256, k % 512 - 256], axis=1)
