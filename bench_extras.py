"""bench_extras.py -- the legs of the bench that are NOT the timed headline region: the CPU baseline + parity check of path A
(the only place outside tests/ and smoke() that calls oracle/), the ICP legs (configs[2], hard list, kinfu-like list, all pairs),
the FragmentOptimizer figure, and the configs[3] / configs[4] child runs.  bench.py imports these; everything they return goes to
bench_full.json, the compact stdout line carries a handful of their numbers (bench.compact_line)."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
BENCH_PY = os.path.join(ROOT, "bench.py")
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec

class _StdoutToStderr:
    """The reference's own code prints to stdout (cout / printf); keep stdout for the ONE JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def cpu_baseline(sc, depth_host, n_sample, files_dir):
    with _StdoutToStderr():
        return _cpu_baseline(sc, depth_host, n_sample, files_dir)


def _cpu_baseline(sc, depth_host, n_sample, files_dir):
    """Reference CPU path on the host cores of this box, same frames, same files-based setup as Integrate.exe (pose.log,
    seg.log and g.ctr are written to files_dir; the GPU leg of the parity check reads the SAME files, so both sides see the
    same text-rounded poses and lattices).
    Returns (json object, {unit key: (sdf_, weight_)} of the volume the timed run left behind) -- the volume is what
    bench.py's parity check compares the GPU volume of the same frames with."""
    import numpy as np
    from elasticreconstruction_amd import formats
    from oracle import pyoracle
    interval = sc["interval"]
    num = n_sample // interval
    if pyoracle.have_ref():
        def run_ref(uncapped, keep_volume):
            d = files_dir
            if True:
                pose = [formats.FramedTransformation(i, i, i + 1, sc["pose"][i]) for i in range(num)]
                seg = [formats.FramedTransformation(i, i, i + 1, sc["seg"][i]) for i in range(n_sample)]
                # one extra fragment of entries so that frame n_sample is still integrated (reference off-by-one)
                formats.save_log(os.path.join(d, "pose.log"), pose + [formats.FramedTransformation(num, num, num + 1, sc["pose"][num - 1])])
                formats.save_log(os.path.join(d, "seg.log"), seg + [formats.FramedTransformation(n_sample + j, n_sample + j, n_sample + j + 1,
                                                                                                sc["seg"][n_sample - 1]) for j in range(interval)])
                formats.save_ctr(os.path.join(d, "g.ctr"), sc["grids"][:num])
                ref = pyoracle.RefApp(uncapped=uncapped)
                ref.init(pose_traj=os.path.join(d, "pose.log"), seg_traj=os.path.join(d, "seg.log"), ctr=os.path.join(d, "g.ctr"),
                         num=num, resolution=sc["resolution"], length=sc["length"], interval=interval)
                t0 = time.perf_counter()
                for f in range(n_sample):
                    ref.execute(f + 1, depth_host[f])
                dt = time.perf_counter() - t0
                vol = {int(k): ref.read_unit(int(k)) for k in ref.unit_keys()} if keep_volume else None
                ref.close()
            return n_sample / dt, vol
        as_written, vol = run_ref(False, True)
        out = {"value": as_written, "unit": "frames/s", "cores": 8, "kind": "reference",
               "sample_short": "first %d frames, reference Execute(), num_threads(8)" % n_sample,
               "sample": "first %d frames of the same stream through the reference's own CIntegrateApp::Execute "
                         "(Reproject+ScaleDepth+Integrate), compiled unmodified, num_threads( 8 ) as hard-coded; "
                         "%d host hardware threads present" % (n_sample, os.cpu_count() or 0)}
        try:
            # same sources with the num_threads clause erased at build time: OpenMP picks the thread count
            out["uncapped"] = {"value": run_ref(True, False)[0], "unit": "frames/s",
                               "threads": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))}
        except Exception as ex:
            out["uncapped"] = {"value": None, "note": str(ex)}
        return out, vol
    from elasticreconstruction_amd import synth
    ora = pyoracle.OracleVolume()
    warp = synth.warp_arrays(sc, 0, n_sample)
    t0 = time.perf_counter()
    for f in range(n_sample):
        dd = ora.Reproject(depth_host[f], sc["grids"][f // interval], sc["resolution"], sc["length"], warp["seg"][f], warp["madj"][f])
        ora.Integrate(dd, sc["traj"][f])
    dt = time.perf_counter() - t0
    vol = {int(k): ora.read_unit(int(k)) for k in ora.unit_keys()}
    return {"value": n_sample / dt, "unit": "frames/s", "cores": os.cpu_count() or 1, "kind": "port",
            "sample_short": "first %d frames, oracle/tsdf_oracle.c" % n_sample,
            "sample": "first %d frames of the same stream through oracle/tsdf_oracle.c (oracle/_ref not present)" % n_sample}, vol


def parity_check(vol, ref_units, n_frames, kind):
    """GPU volume vs the CPU volume of the SAME frames (the one cpu_baseline just produced): identical unit key sets and
    identical float bit patterns of every sdf_ / weight_ array (path A's bar is bit-exact)."""
    import hashlib
    import numpy as np
    kg = [int(k) for k in vol.unit_keys()]
    kr = sorted(ref_units.keys())
    res = {"frames": n_frames, "against": "reference build (oracle/_ref)" if kind == "reference" else "oracle port",
           "units_gpu": len(kg), "units_cpu": len(kr), "keys_equal": kg == kr}
    bad, hg, hr, sw = 0, hashlib.sha256(), hashlib.sha256(), 0.0
    if res["keys_equal"]:
        for k in kr:
            sg, wg = vol.read_unit(k)
            so, wo = ref_units[k]
            hg.update(sg.tobytes()); hg.update(wg.tobytes())
            hr.update(np.ascontiguousarray(so, np.float32).tobytes()); hr.update(np.ascontiguousarray(wo, np.float32).tobytes())
            if not (np.array_equal(wg, wo) and np.array_equal(sg.view(np.uint32), np.asarray(so, np.float32).view(np.uint32))):
                bad += 1
            sw += float(wo.sum(dtype=np.float64))
    res.update({"units_differing": bad, "sha256_gpu": hg.hexdigest()[:16], "sha256_cpu": hr.hexdigest()[:16], "sum_weight": sw,
                "bit_exact": bool(res["keys_equal"] and bad == 0)})
    return res


def sampled_parity(sc, depth, n_sample, interval, max_units, device):
    """configs[3] / configs[4] (strong-scaling jobs of 10 000 / 5000 frames): the CPU leg and the parity check on a SAMPLED stream --
    runs of 10 consecutive frames spread from the first to the last fragment of the job, every frame with its true frame id (its own
    lattice of the job's .ctr, its own trajectory entry) -- through the reference's own CIntegrateApp::Execute (oracle/_ref) and
    through the host mirror on the GPU, both reading the same pose.log / seg.log / g.ctr.  The head of such a job says nothing about
    its hard part (negative unit coordinates, > 512 hashed units appear as the path drifts outward).
    Returns (cpu_baseline object, parity_checked object)."""
    import numpy as np
    import torch
    from elasticreconstruction_amd import synth
    from elasticreconstruction_amd.tsdf import IntegrateApp
    from oracle import pyoracle, refcheck
    if not pyoracle.have_ref():
        return {"value": None, "kind": "reference", "note": "oracle/_ref not present on this host"}, None
    run_len = min(10, interval)
    ids = refcheck.sampled_frames(sc["n"], interval, max(2, n_sample // run_len), run_len)
    host = synth.to_numpy_u16(depth.view(torch.int16)[torch.as_tensor(ids, device=depth.device)])
    with tempfile.TemporaryDirectory() as fdir:
        with _StdoutToStderr():
            ref_units, dt, paths = refcheck.reference_volume_of_frames(sc, host, ids, fdir)
        app = IntegrateApp(max_units=max_units, device=device)
        app.pose_filename_, app.seg_filename_, app.ctr_filename_ = paths
        app.ctr_num_, app.ctr_resolution_, app.ctr_length_, app.ctr_interval_ = sc["n"] // interval, sc["resolution"], sc["length"], interval
        app.Init()
        for k, f in enumerate(ids):
            app.Execute(int(f) + 1, host[k])
        app.Finish(save=False)
    par = parity_check(app.volume_, ref_units, len(ids), "reference")
    c = refcheck.unit_coordinates(sorted(ref_units))
    par.update({"frames_sampled": "%d runs of %d consecutive frames, first fragment to last, true frame ids" % (len(ids) // run_len, run_len),
                "job_frames": sc["n"], "lattices_in_ctr": sc["n"] // interval,
                "unit_coordinate_min": [int(x) for x in c.min(0)], "unit_coordinate_max": [int(x) for x in c.max(0)],
                "units_at_negative_coordinates": int((c < 0).any(axis=1).sum()),
                "units_outside_the_512_cube_region": int(((c < 0) | (c > 7)).any(axis=1).sum())})
    app.volume_.close()
    cpu = {"value": len(ids) / dt, "unit": "frames/s", "cores": 8, "kind": "reference",
           "sample_short": "%d frames sampled across the job, reference Execute(), num_threads(8)" % len(ids),
           "sample": "%d frames sampled across the %d-frame job (runs of %d) through the reference's own CIntegrateApp::Execute "
                     "(Reproject+ScaleDepth+Integrate), compiled unmodified, num_threads( 8 ) as hard-coded; %d host hardware threads present"
                     % (len(ids), sc["n"], run_len, os.cpu_count() or 0)}
    return cpu, par


def icp_section(n_pairs, device, with_cpu=True, n_frag=25):
    """configs[2] shape: n_pairs fragment pairs over n_frag DISTINCT fragments of 250 k points each (seeded surfels of the
    synthetic room seen from n_frag places on the config-2 circle; pair k = fragment a with its 1st / 2nd neighbour, ground
    truth o perturbation (<= 2 deg, 2 cm) as the initial guess) through the reference flow: inlier pre-check + ICP (<= 20
    iterations) + FindCorrespondence + information matrix.  Secondary metric (pairs/s).  CPU row: the reference's own CCorresApp
    (compiled in place, PCL replaced by oracle/stub_corres) on the first 8 pairs -- one per thread of its num_threads( 8 ) loops --
    plus the restatement (oracle/icp_oracle.cpp) on the same 8 pairs as the parity check."""
    import numpy as np
    from elasticreconstruction_amd import synth
    from elasticreconstruction_amd.icp import Cloud, count_inliers, find_correspondence, icp_align
    frs = synth.fragment_set(n_frag, 250000, device="cuda:%d" % device)
    clouds, hosts, build_ms = [], [], []
    Cloud(frs[0][0][:1000], frs[0][1][:1000], 0.03, device).close()     # first-use costs (module load) stay out of the figure
    for x, n, F in frs:
        t0 = time.perf_counter()
        clouds.append((Cloud(x, n, 0.03, device), F))
        build_ms.append((time.perf_counter() - t0) * 1e3)
        hosts.append((x, n))
    # the same fragments through er_cloud_create_batch from PAGE-LOCKED arrays (what a host that reads its PCD files into er_host_alloc memory
    # gets): uploads back to back on a copy stream, grids underneath -- the list is PCIe-bound (24 bytes per point)
    from elasticreconstruction_amd import _ffi
    arena = _ffi.PinnedArena()
    arena.reset(sum(x.nbytes + n.nbytes for x, n in hosts) + 16384 * len(hosts))
    pinned = []
    t0 = time.perf_counter()
    for x, n in hosts:
        px, pn = arena.take(x.shape, np.float32), arena.take(n.shape, np.float32)
        px[...] = x
        pn[...] = n
        pinned.append((px, pn))
    staging_s = time.perf_counter() - t0                      # pageable -> page-locked staging of the whole fragment list (reported)
    batch_s = []
    for _ in range(6):
        t0 = time.perf_counter()
        tmp = Cloud.create_batch(pinned, 0.03, device)
        batch_s.append(time.perf_counter() - t0)
        for c in tmp:
            c.close()
        time.sleep(0.05)     # hipFree returns before the runtime has finished releasing the memory: a build that starts right behind the previous
                             # batch's destruction takes 6.2 ms instead of 3.8 ms (profiles/r04B_cloud_build_after_free.txt); a job builds its clouds once
    arena.close()
    batch_build_s = float(np.median(batch_s[2:]))
    frs_host = [(x, n, F) for (x, n), (_, F) in zip(hosts, clouds)]
    pairs = synth.config2_pair_list(frs_host, n_pairs)

    from elasticreconstruction_amd.icp import count_inliers_batch, find_correspondence_batch, icp_align_batch

    def run_pair(a, b, T):
        tgt, src = clouds[a][0], clouds[b][0]
        cnt = count_inliers(src, tgt, T, 0.03)
        fin, iters, conv, _ = icp_align(src, tgt, T.astype(np.float32), 0.03, 20, 1e-6, 0)
        corr, info = find_correspondence(src, tgt, fin.astype(np.float64), 0.015, 0.8660, True)
        return cnt, iters, corr.shape[0]

    phase = [0.0, 0.0, 0.0]

    last = {}

    def run_list(plist, cl=None):
        """The reference's flow over a pair list: Registration loop (pre-check + ICP), then the FindCorrespondence loop."""
        cl = clouds if cl is None else cl
        srcs, tgts = [cl[b][0] for _, b, _ in plist], [cl[a][0] for a, _, _ in plist]
        t0 = time.perf_counter()
        cnts = count_inliers_batch(srcs, tgts, [T for _, _, T in plist], 0.03)
        t1 = time.perf_counter()
        fins, iters, conv, _ = icp_align_batch(srcs, tgts, [T.astype(np.float32) for _, _, T in plist], 0.03, 20, 1e-6, 0)
        t2 = time.perf_counter()
        lists, infos = find_correspondence_batch(srcs, tgts, [F.astype(np.float64) for F in fins], 0.015, 0.8660, True, copy=False)
        t3 = time.perf_counter()
        phase[:] = [(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3]
        last.update(conv=conv, infos=infos)
        return cnts, iters, [l.shape[0] for l in lists], fins, lists

    run_pair(*pairs[0])
    run_list(pairs)          # steady state: workspaces, the page-locked result arena and the caches are warm
    t0 = time.perf_counter()
    its1 = 0
    nseq = min(n_pairs, 8)
    for p in pairs[:nseq]:
        its1 += run_pair(*p)[1]
    dt1 = time.perf_counter() - t0
    # the single-pair entry points the way the reference's loops would call them: "#pragma omp parallel for num_threads( 8 )
    # schedule( dynamic )" over the pair list (CorresApp.cpp:121,220) -- 8 host threads, each pair three blocking calls on a
    # workspace of its own (ctypes releases the GIL inside the calls)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(lambda p: run_pair(*p), pairs[:8]))        # warm: one workspace per thread
        t0 = time.perf_counter()
        list(ex.map(lambda p: run_pair(*p), pairs))
        dt8 = time.perf_counter() - t0
    # the whole list takes ~10 ms, the same order as one scheduling hiccup on a shared host: median of 7 passes
    dts, phases = [], []
    for _ in range(4):       # (untimed: after the single-call runs above the first list passes take 10, 10, 6, 6 ms before they settle at
        run_list(pairs)      #  5.3 ms -- clocks and the list workspace's page-locked arena warm up; every pass is reported in pass_ms)
    for _ in range(7):
        t0 = time.perf_counter()
        cnts, iters, ncs, fins, lists = run_list(pairs)
        dts.append(time.perf_counter() - t0)
        phases.append(list(phase))
    order = sorted(range(len(dts)), key=lambda q: dts[q])
    dt = dts[order[len(dts) // 2]]
    phase = phases[order[len(dts) // 2]]
    its, ncor = int(np.sum(iters)), int(np.sum(ncs))
    ncpu = min(n_pairs, 8)                                   # pairs that go to the CPU legs (>= 5; 8 = one per thread of the reference's num_threads( 8 ))
    head_lists = [np.array(l) for l in lists[:ncpu]]         # (views into the page-locked arena: copy before it is reused)
    # the fused entry point (Registration + FindCorrespondence in one call, the shares of the list on host threads of their own)
    from elasticreconstruction_amd.icp import registration_batch
    f_srcs, f_tgts, f_T = [clouds[b][0] for _, b, _ in pairs], [clouds[a][0] for a, _, _ in pairs], [T for _, _, T in pairs]
    fdt, fused = [], None
    for _ in range(9):
        t0 = time.perf_counter()
        fused = registration_batch(f_srcs, f_tgts, f_T, 0.03, 40000, 0.25, 20, 1e-6, 0, 0.015, 0.8660, want_info=True, copy=False)
        fdt.append(time.perf_counter() - t0)
    fused_same = bool(all(np.array_equal(fused["T"][k], fins[k]) for k in range(n_pairs)) and [len(l) for l in fused["lists"]] == [int(c) for c in ncs])
    fdt = fdt[2:]
    npts = sum(len(c[0]) for c in clouds) / float(len(clouds))
    # B_B = P [ (I+2) 12 + I_c 24 ] + C 24 per pair (P as the upper bound of the in-range points); NN traversal excluded
    bb = float(sum(len(clouds[b][0]) * ((int(i) + 2) * 12 + int(i) * 24) + int(c) * 24 for (_, b, _), i, c in zip(pairs, iters, ncs)))
    bb_pre = float(sum(len(clouds[b][0]) * 12 for _, b, _ in pairs))
    bb_icp = float(sum(len(clouds[b][0]) * int(i) * 36 for (_, b, _), i in zip(pairs, iters)))
    bb_fc = float(sum(len(clouds[b][0]) * 12 + int(c) * 24 for (_, b, _), c in zip(pairs, ncs)))
    build_total = float(np.sum(build_ms)) * 1e-3
    res = {"pairs_per_s": n_pairs / dt, "pairs": n_pairs, "distinct_fragments": len(clouds), "points_per_fragment": npts,
           "mean_icp_iterations": its / n_pairs,
           "roofline": {"bound": "hbm (lower bound: NN-structure traversal bytes are implementation-defined and excluded)",
                        "algorithmic_bytes_per_pair": bb / n_pairs, "achieved": bb / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": bb / dt / 1e9 / HBM_PEAK_GBS,
                        # SURVEY.md 8d's B_B split by the phase (= kernel family) that moves each term, each over ITS OWN wall time of the median pass
                        "phases": {ph: {"kernels": kn, "algorithmic_bytes_per_list": b, "ms": ms, "achieved": b / (ms * 1e-3) / 1e9,
                                        "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                   for ph, kn, b, ms in (("pre_check", "k_count_inliers", bb_pre, phase[0]),
                                                         ("icp", "k_icp_iter + k_icp_final", bb_icp, phase[1]),
                                                         ("find_correspondence", "k_find_corr + k_count_blocks / k_scan_blocks / k_compact + the list copies to the host",
                                                          bb_fc, phase[2]))},
                        "phases_definition": "B_B = P [(I + 2) 12 + I 24] + C 24 per pair: the pre-check reads P x 12, the ICP loop P x I x (12 + 24), "
                                             "FindCorrespondence P x 12 + C x 24; the three sum to algorithmic_bytes_per_pair x pairs"},
           "cloud_build_ms": {"per_fragment_median": float(np.median(build_ms)), "per_fragment_max": float(np.max(build_ms)),
                              "what": "er_cloud_create: upload + uniform-grid build of one fragment, once per fragment (the reference "
                                      "rebuilds a kd-tree per pair and per function, CorresApp.cpp:129,238)",
                              "batch_all_fragments_ms": 1e3 * batch_build_s, "batch_per_fragment_ms": 1e3 * batch_build_s / len(clouds),
                              "batch_input_GB_per_s": sum(x.nbytes + n.nbytes for x, n in hosts) / batch_build_s / 1e9,
                              "batch_what": "er_cloud_create_batch over all %d fragments from page-locked host arrays: all uploads queued up front on two copy streams, "
                                            "chunks of up to 8 clouds share one set of grid launches, one host wait per chunk (PCIe-bound: 24 bytes per point); "
                                            "median of 4 calls, each 50 ms after the previous batch was destroyed" % len(clouds)},
           # this key is back on its round 1-3 definition (clouds built ONE BY ONE from pageable arrays, er_cloud_create);
           # round 4 had silently moved it to the batched build from page-locked arrays, which now has a key of its own
           "pairs_per_s_incl_cloud_build": n_pairs / (dt + build_total),
           "pairs_per_s_incl_cloud_build_batched": n_pairs / (dt + batch_build_s),
           "pairs_per_s_incl_cloud_build_batched_and_staging": n_pairs / (dt + batch_build_s + staging_s),
           "pairs_per_s_incl_cloud_build_basis": "pairs_per_s_incl_cloud_build = list time + the sum of 25 er_cloud_create calls from pageable arrays (rounds 1-3 and "
                                                 "again now; BENCH_r04's value under this key was the batched figure); _batched = list time + ONE er_cloud_create_batch from "
                                                 "page-locked arrays (median of the last 4 of 6 calls, 50 ms after the previous batch was freed); _batched_and_staging adds the "
                                                 "pageable -> page-locked copy of the 25 fragments (%.2f ms), which a host that reads its PCD files straight into er_host_alloc "
                                                 "memory does not pay" % (1e3 * staging_s),
           "mean_correspondences": ncor / n_pairs, "nn_queries_per_s": npts * (its + 2 * n_pairs) / dt,
           "flow": "er_icp_count_inliers_batch + er_icp_align_batch, then er_find_correspondence_batch over the pair list",
           "phase_ms": {"pre_check": phase[0], "icp": phase[1], "find_correspondence": phase[2]},
           "timing": "median of 7 passes over the pair list; min %.2f ms, max %.2f ms per pass" % (min(dts) * 1e3, max(dts) * 1e3),
           "pass_ms": [round(1e3 * t, 3) for t in dts],
           "fused_entry": {"pairs_per_s": n_pairs / float(np.median(fdt)), "pass_ms": [round(1e3 * t, 3) for t in fdt], "accepted": int(fused["accepted"].sum()),
                           "equals_the_three_calls": fused_same,
                           "what": "er_registration_batch: pre-check, accept rule, ICP and FindCorrespondence of the whole list in ONE call, the list cut into "
                                   "up to 6 shares of >= 8 pairs that run on host threads / workspaces of their own (ER_ICP_SHARES)"},
           "single_call_pairs_per_s": nseq / dt1,
           "single_call_8_host_threads_pairs_per_s": n_pairs / dt8}
    res["_pass_s"] = dt
    # ---- the device-resident hand-off: the same fused call with list buffers that live in HBM (the 83 MB of lists never cross
    # PCIe), consumed where they are by er_fopt_set_correspondences_dev (the sort by lattice cell pair on the GPU) ----
    try:
        from elasticreconstruction_amd.fopt import FragmentOptimizer
        from elasticreconstruction_amd.icp import DeviceLists, registration_batch_dev
        dl = DeviceLists(f_srcs, device)
        ddt = []
        for _ in range(9):
            t0 = time.perf_counter()
            dres = registration_batch_dev(f_srcs, f_tgts, f_T, dl, 0.03, 40000, 0.25, 20, 1e-6, 0, 0.015, 0.8660, want_info=True)
            ddt.append(time.perf_counter() - t0)
        ddt = ddt[2:]
        same = bool(np.array_equal(dres["T"], fused["T"]) and [int(c) for c in dl.counts] == [len(l) for l in fused["lists"]] and
                    all(np.array_equal(dl.download(k), np.asarray(fused["lists"][k])) for k in (0, n_pairs // 2, n_pairs - 1)))
        fo = FragmentOptimizer(len(clouds), 8, 3.0, device)
        for f, (x, n) in enumerate(hosts):
            fo.SetCloud(f, x, n)
        ids = [(a, b) for a, b, _ in pairs]
        t0 = time.perf_counter()
        ng_dev = fo.SetCorrespondencesDev(ids, dl)
        t_dev = time.perf_counter() - t0
        t0 = time.perf_counter()
        ng_dev = fo.SetCorrespondencesDev(ids, dl)
        t_dev = min(t_dev, time.perf_counter() - t0)
        host_lists = [(a, b, np.array(fused["lists"][k])) for k, (a, b) in enumerate(ids)]
        t0 = time.perf_counter()
        ng_host = fo.SetCorrespondences(host_lists)
        t_host = time.perf_counter() - t0
        fo.close()
        res["device_hand_off"] = {"pairs_per_s": n_pairs / float(np.median(ddt)), "list_ms": 1e3 * float(np.median(ddt)), "pass_ms": [round(1e3 * t, 3) for t in ddt],
                                  "equals_the_host_path": same, "list_bytes_kept_in_hbm": int(dl.counts.sum()) * 8,
                                  "fopt_set_correspondences_ms": {"device_lists": 1e3 * t_dev, "host_lists": 1e3 * t_host, "groups": int(ng_dev),
                                                                  "groups_equal": bool(ng_dev == ng_host), "correspondences": int(dl.counts.sum())},
                                  "what": "er_registration_batch with list buffers in HBM (er_device_alloc; the list copies are device-to-device), then "
                                          "er_fopt_set_correspondences_dev on them (keys + ONE radix sort + run lengths on the GPU) against er_fopt_set_correspondences "
                                          "on the host copies (a stable_sort per list on one host core, then three uploads)"}
        dl.close()
    except Exception as ex:
        res["device_hand_off"] = {"error": repr(ex)[:400]}
    # RansacCurvature::getFitness over a hypothesis list (down-sampled source, as GlobalRegistration uses it)
    from elasticreconstruction_amd.icp import ransac_fitness_batch
    sub = np.sort(np.random.default_rng(11).choice(hosts[1][0].shape[0], 5000, replace=False))
    small = Cloud(hosts[1][0][sub], hosts[1][1][sub], 0.03, device)
    base = np.linalg.inv(clouds[0][1]) @ clouds[1][1]
    H = np.stack([(base @ synth.perturbation(900 + k, 3.0, 0.05)).astype(np.float32) for k in range(256)])
    H = np.tile(H, (64, 1, 1))
    ransac_fitness_batch(small, clouds[0][0], H[:256], 0.03)
    t0 = time.perf_counter()
    ransac_fitness_batch(small, clouds[0][0], H, 0.03)
    dth = time.perf_counter() - t0
    res["ransac_fitness"] = {"hypotheses_per_s": H.shape[0] / dth, "hypotheses": int(H.shape[0]), "source_points": 5000,
                             "target_points": len(clouds[0][0]), "what": "er_ransac_fitness_batch = RansacCurvature::getFitness per hypothesis"}
    # ---- a HARD pair list: the same fragments, guesses up to 6 deg / 6 cm off (three times the configs[2] perturbation),
    # so the 20-iteration budget, the transform criterion and the iteration limit are all on the timed path ----
    hard = synth.hard_pair_list(frs_host, n_pairs)
    run_list(hard)
    hd = []
    for _ in range(3):
        t0 = time.perf_counter()
        h_cnts, h_iters, _, h_fins, h_lists = run_list(hard)
        hd.append(time.perf_counter() - t0)
    h_conv, h_infos = last["conv"], last["infos"]
    h_err = [float(np.abs(F.astype(np.float64) - np.linalg.inv(clouds[a][1]) @ clouds[b][1]).max()) for F, (a, b, _) in zip(h_fins, hard)]
    res["hard_set"] = {"pairs_per_s": n_pairs / float(np.median(hd)), "mean_icp_iterations": float(np.mean(h_iters)), "max_icp_iterations": int(np.max(h_iters)),
                       "pairs_at_the_iteration_limit": int(np.sum(np.asarray(h_iters) >= 20)), "converged": int(np.sum(h_conv)),
                       "guess": "ground truth o perturbation of <= 6 deg / 6 cm (synth.hard_pair_list)", "max_abs_T_error_vs_ground_truth": max(h_err),
                       "pairs_within_2mm_of_ground_truth": int(np.sum(np.asarray(h_err) < 2e-3)),
                       "nn_queries_per_s": npts * (int(np.sum(h_iters)) + 2 * n_pairs) / float(np.median(hd))}
    if with_cpu:
        # >= 8 pairs of the hard list -- every pair at the iteration limit (<= 3), the one that ends farthest from the ground truth, the slowest
        # converging one, then the first ones -- against the reference's own compiled CCorresApp (the 20-iteration, transform-
        # criterion and MSE exits at 250 k points were timed but never compared)
        try:
            from oracle import refcheck
            from oracle.pyoracle import RefCorres
            if RefCorres.available():
                sel = refcheck.select_hard(h_iters, h_err, want=8)
                h_lists_c = {k: np.array(h_lists[k]) for k in sel}
                with tempfile.TemporaryDirectory() as hdir, _StdoutToStderr():
                    chk = refcheck.check_pairs_against_reference(frs_host, hard, sel, h_cnts, h_fins, h_iters, h_conv, h_lists_c, h_infos, hdir)
                chk["ok"] = True
                res["hard_set"]["parity_checked_reference"] = chk
            else:
                res["hard_set"]["parity_checked_reference"] = {"ok": None, "note": "oracle/_ref/libref_corres.so did not travel"}
        except AssertionError as ex:
            res["hard_set"]["parity_checked_reference"] = {"ok": False, "mismatch": str(ex)[:400]}
        except Exception as ex:
            res["hard_set"]["parity_checked_reference"] = {"ok": None, "note": "checker failed to run: %s" % ex}
    # ---- fragments that look like fragments: synth.kinfu_fragment -- 50 depth frames of a hand-held sweep through THIS library's
    # Integrate path, zero crossings of the volume, TSDF-gradient normals with NaNs at the border of the observed region (filtered like LoadData does),
    # thinned ~ 1 / z^2, odd fragments from depth images with 2 mm noise -- through the same flow, same list shape (guesses <= 2 deg / 2 cm) ----
    # (runs AFTER the hard list's reference check: h_lists are views into the page-locked result arena, which every run_list call reuses)
    try:
        kfr, kst = [], []
        for i in range(n_frag):
            x, n, F, st = synth.kinfu_fragment(i, 2 * n_frag, 250000, noise_mm=2.0 if i % 2 else 0.0, device=device)   # sweeps 7.2 degrees apart
            ok = ~np.isnan(n).any(axis=1)
            kfr.append((np.ascontiguousarray(x[ok]), np.ascontiguousarray(n[ok]), F))
            kst.append(st)
        kcl = [(Cloud(x, n, 0.03, device), F) for x, n, F in kfr]
        kpairs = synth.chain_pair_list(kfr, n_pairs, 2.0, 0.02, 700)          # (the sweeps cover half a circle: neighbours 1 / 2 / 3 apart, no wrap-around)
        run_list(kpairs, kcl)
        run_list(kpairs, kcl)
        kd, kph = [], []
        for _ in range(5):
            t0 = time.perf_counter()
            k_cnts, k_iters, k_ncs, k_fins, k_lists = run_list(kpairs, kcl)
            kd.append(time.perf_counter() - t0)
            kph.append(list(phase))
        k_conv, k_infos = last["conv"], last["infos"]
        ko = sorted(range(len(kd)), key=lambda q: kd[q])[len(kd) // 2]
        occ = [synth.cell_occupancy(x) for x, _, _ in kfr]
        uocc = [synth.cell_occupancy(x) for x, _ in hosts[:4]]
        k_err = [float(np.abs(F.astype(np.float64) - np.linalg.inv(kfr[a][2]) @ kfr[b][2]).max()) for F, (a, b, _) in zip(k_fins, kpairs)]
        knpts = float(np.mean([len(x) for x, _, _ in kfr]))
        res["realistic"] = {"pairs_per_s": n_pairs / kd[ko], "pairs": n_pairs, "ratio_to_the_uniform_list": (n_pairs / kd[ko]) / (n_pairs / dt),
                            "phase_ms": {"pre_check": kph[ko][0], "icp": kph[ko][1], "find_correspondence": kph[ko][2]},
                            "pass_ms": [round(1e3 * t, 3) for t in kd],
                            "points_per_fragment_after_nan_filter": knpts, "nan_normal_fraction": float(np.mean([st["nan_fraction"] for st in kst])),
                            "zero_crossings_per_fragment": float(np.mean([st["zero_crossings"] for st in kst])),
                            "mean_icp_iterations": float(np.mean(k_iters)), "max_icp_iterations": int(np.max(k_iters)), "converged": int(np.sum(k_conv)),
                            "mean_correspondences": float(np.mean(k_ncs)),
                            "nn_queries_per_s": knpts * (int(np.sum(k_iters)) + 2 * n_pairs) / kd[ko],
                            "cell_occupancy": {"max": max(o[0] for o in occ), "mean": float(np.mean([o[1] for o in occ])),
                                               "uniform_list_max": max(o[0] for o in uocc), "uniform_list_mean": float(np.mean([o[1] for o in uocc])),
                                               "what": "points per occupied 3 cm cell of the target grids"},
                            "ground_truth_error": {"median": float(np.median(k_err)), "max": max(k_err)},
                            "what": "the configs[2] flow on 25 kinfu-like fragments (synth.kinfu_fragment: TSDF zero crossings of a 50-frame sweep, gradient "
                                    "normals, NaN filter, ~1/z^2 thinning, 2 mm depth noise on the odd fragments); tests/test_icp_gpu.py::"
                                    "test_kinfu_like_fragments_at_config2_size checks all 50 pairs of this list against the oracle and its hard variant against the "
                                    "reference's CCorresApp"}
        if with_cpu:
            from oracle import refcheck
            from oracle.pyoracle import RefCorres
            if RefCorres.available():
                try:
                    sel = sorted({0, int(np.argmax(k_iters)), int(np.argmax(k_err)), n_pairs - 1})
                    k_lists_c = {k: np.array(k_lists[k]) for k in sel}
                    with tempfile.TemporaryDirectory() as kdir, _StdoutToStderr():
                        chk = refcheck.check_pairs_against_reference(kfr, kpairs, sel, k_cnts, k_fins, k_iters, k_conv, k_lists_c, k_infos, kdir)
                    chk["ok"] = True
                    res["realistic"]["parity_checked_reference"] = chk
                except AssertionError as ex:
                    res["realistic"]["parity_checked_reference"] = {"ok": False, "mismatch": str(ex)[:400]}
        for c, _ in kcl:
            c.close()
    except Exception as ex:                                            # the leg is additional evidence: never lose the line over it
        res["realistic"] = {"error": repr(ex)[:400]}
    if not with_cpu:
        return res
    try:
        from oracle.pyoracle import IcpOracle, RefCorres
        need = sorted({q for a, b, _ in pairs[:ncpu] for q in (a, b)} | {0, 1})
        if RefCorres.available():
            # the reference's OWN CCorresApp::Registration + FindCorrespondence (BuildCorrespondence/CorresApp.cpp compiled in place,
            # PCL replaced by oracle/stub_corres: exact kd-tree + the PCL 1.7 ICP restated on the reference's vendored Eigen)
            def run_ref(uncapped):
                app = RefCorres(reg_dist=0.03, uncapped=uncapped)
                idx = {q: app.add_cloud(*hosts[q]) for q in need}
                for a, b, T in pairs[:ncpu]:
                    app.add_pair(idx[a], idx[b], len(need), T)
                with _StdoutToStderr():
                    t0 = time.perf_counter()
                    app.Registration()
                    app.FindCorrespondence()
                    dt_ref = time.perf_counter() - t0
                out = app.pairs()
                app.close()
                return ncpu / dt_ref, out
            v8, ref_pairs = run_ref(False)
            res["cpu_baseline"] = {"value": v8, "unit": "pairs/s", "cores": 8, "kind": "reference", "pairs": ncpu,
                                   "sample": "the first %d pairs of the same list through the reference's own CCorresApp::Registration + "
                                             "FindCorrespondence (compiled unmodified, num_threads( 8 ) as hard-coded; PCL = oracle/stub_corres); "
                                             "%d host hardware threads present" % (ncpu, os.cpu_count() or 0)}
            try:
                res["cpu_baseline"]["uncapped"] = {"value": run_ref(True)[0], "unit": "pairs/s",
                                                   "threads": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))}
            except Exception as ex:
                res["cpu_baseline"]["uncapped"] = {"value": None, "note": str(ex)}
            # HIP vs the reference program's members on the bench's own pairs: frame_ (= correspondences after FindCorrespondence)
            # within the handful of borderline points a 1e-7 difference in T can flip, T within 1e-5, information to 1e-3 of its scale
            worst_T, worst_n, ok = 0.0, 0, True
            for k, (_, _, frame, T, info) in enumerate(ref_pairs):
                worst_T = max(worst_T, float(np.abs(T - fins[k].astype(np.float64)).max()))
                worst_n = max(worst_n, abs(int(frame) - int(ncs[k])))
                ok = ok and frame != -1
            res["parity_checked_reference"] = {"pairs": ncpu, "against": "the reference's CCorresApp compiled in place (oracle/_ref/libref_corres.so)",
                                               "max_abs_T_diff": worst_T, "tolerance_T": 1e-5, "max_correspondence_count_diff": worst_n,
                                               "ok": bool(ok and worst_T <= 1e-5 and worst_n <= max(3, int(np.max(ncs)) // 1000))}
        oc = {i: IcpOracle(*hosts[i], 0.03) for i in need}
        t0 = time.perf_counter()
        ok, worst = True, 0.0
        for k, (a, b, T) in enumerate(pairs[:ncpu]):
            c_o = oc[b].count_inliers(oc[a], T, 0.03)
            fin, it_o, _, _ = oc[b].align(oc[a], T.astype(np.float32))
            l_o, _ = oc[b].find_correspondence(oc[a], fins[k].astype(np.float64), 0.015, 0.8660, True)
            # HIP vs the CPU restatement on the bench's own pairs: integers and index lists exact, T within 1e-5
            worst = max(worst, float(np.abs(fin - fins[k]).max()))
            ok = ok and int(c_o) == int(cnts[k]) and int(it_o) == int(iters[k]) and np.array_equal(np.asarray(l_o), head_lists[k])
        res["cpu_port_pairs_per_s"] = ncpu / (time.perf_counter() - t0)
        res["parity_checked"] = {"pairs": ncpu, "against": "oracle/icp_oracle.cpp (pinned to the reference's compiled CorresApp by tests/test_corres_reference.py; "
                                                           "the PCL calls inside stay a restatement)", "counts_iterations_lists_exact": bool(ok),
                                 "max_abs_T_diff": worst, "tolerance_T": 1e-5, "ok": bool(ok and worst <= 1e-5)}
        osm = IcpOracle(hosts[1][0][sub], hosts[1][1][sub], 0.03)
        t0 = time.perf_counter()
        for k in range(32):
            osm.ransac_fitness(oc[0], H[k], 0.03)
        res["ransac_fitness"]["cpu_port_hypotheses_per_s"] = 32 / (time.perf_counter() - t0)
        res["cpu_port_note"] = "oracle/icp_oracle.cpp (PCL 1.7 restatement, OpenMP NN over %d threads), %d pairs" % (os.cpu_count() or 1, ncpu)
    except Exception as ex:                                            # the checker is optional for the bench
        res["cpu_port_note"] = "oracle not available: %s" % ex
    return res


def allpairs_section(n_frag, device, rank=0, world=1, with_cpu=True):
    """configs[4] shape, first half: ALL pairs of an n_frag-fragment scene (n_frag (n_frag - 1) / 2 = 4950 for 100) through the
    reference's BuildCorrespondence flow -- Registration pre-check on every pair (CorresApp.cpp:257-281: accepted iff the
    inlier count reaches reg_num_ = 40000 or both ratios exceed 0.25), ICP + FindCorrespondence on the accepted ones --
    with the pairs dealt to the ranks by parallel.pair_shard and no collective.  Fragments: outward-looking views from a
    circle (neighbours overlap, distant ones do not, so the pre-check really rejects most pairs)."""
    import numpy as np
    from elasticreconstruction_amd import parallel, synth
    from elasticreconstruction_amd.icp import Cloud, count_inliers_batch, find_correspondence_batch, icp_align_batch
    frs = synth.fragment_set(n_frag, 250000, radius=0.6, device="cuda:%d" % device)
    clouds = [Cloud(x, n, 0.03, device) for x, n, _ in frs]
    allp = [(i, j) for i in range(n_frag) for j in range(i + 1, n_frag)]
    mine = [allp[p] for p in parallel.pair_shard(len(allp), rank, world)]
    Ts = [np.linalg.inv(frs[i][2]) @ frs[j][2] @ synth.perturbation(9000 + i * n_frag + j, 1.0, 0.01) for i, j in mine]

    last = {}

    def run():
        t0 = time.perf_counter()
        cnts = count_inliers_batch([clouds[j] for _, j in mine], [clouds[i] for i, _ in mine], Ts, 0.03)
        npts = np.array([[len(clouds[i]), len(clouds[j])] for i, j in mine], np.float64)
        acc = (cnts >= 40000) | ((cnts / npts[:, 0] > 0.25) & (cnts / npts[:, 1] > 0.25))
        ai = np.nonzero(acc)[0]
        t1 = time.perf_counter()
        fins, iters, conv, _ = icp_align_batch([clouds[mine[k][1]] for k in ai], [clouds[mine[k][0]] for k in ai],
                                               [Ts[k].astype(np.float32) for k in ai], 0.03, 20, 1e-6, 0)
        t2 = time.perf_counter()
        lists, infos = find_correspondence_batch([clouds[mine[k][1]] for k in ai], [clouds[mine[k][0]] for k in ai],
                                                 [F.astype(np.float64) for F in fins], 0.015, 0.8660, True, copy=False)
        t3 = time.perf_counter()
        last.update(fins=fins, iters=iters, conv=conv, lists=lists, infos=infos)
        return cnts, acc, iters, [l.shape[0] for l in lists], (t1 - t0, t2 - t1, t3 - t2)
    run()
    cnts, acc, iters, ncs, ph = run()
    dt = sum(ph)
    res = {"fragments": n_frag, "pairs_total": len(allp), "pairs_this_rank": len(mine), "accepted_this_rank": int(acc.sum()),
           "rejected_by_pre_check": int((~acc).sum()), "pairs_per_s": len(mine) / dt, "accepted_pairs_per_s": int(acc.sum()) / dt,
           "mean_icp_iterations": float(np.mean(iters)) if len(iters) else 0.0, "mean_correspondences": float(np.mean(ncs)) if ncs else 0.0,
           "phase_ms": {"pre_check_all_pairs": 1e3 * ph[0], "icp_accepted": 1e3 * ph[1], "find_correspondence_accepted": 1e3 * ph[2]},
           "sharding": "pair p -> rank p mod %d, fragments replicated, no collective" % world, "_pass_s": dt}
    if with_cpu:
        # 14 random pairs + the first four accepted ones (24 + 4 until round 4: the leg took 90 s of the default run) against the reference's own compiled CCorresApp (the accept rule of
        # CorresApp.cpp:257-281 decides on BOTH sides which of them get an ICP and a correspondence list); where that build did not
        # travel, the pre-check counts against the restatement
        try:
            from oracle import refcheck
            from oracle.pyoracle import IcpOracle, RefCorres
            rng = np.random.default_rng(5)
            pick = sorted(set(int(k) for k in rng.choice(len(mine), 14, replace=False)) | set(int(k) for k in np.nonzero(acc)[0][:4]))
            if RefCorres.available():
                fins, its, conv, lists, infos = last["fins"], last["iters"], last["conv"], last["lists"], last["infos"]
                pos = {int(k): q for q, k in enumerate(np.nonzero(acc)[0])}
                at = lambda arr, k: arr[pos[k]] if k in pos else None
                plist = [(i, j, T) for (i, j), T in zip(mine, Ts)]
                with tempfile.TemporaryDirectory() as hdir, _StdoutToStderr():
                    chk = refcheck.check_pairs_against_reference(
                        frs, plist, pick, cnts, {k: at(fins, k) for k in pick}, {k: at(its, k) for k in pick}, {k: at(conv, k) for k in pick},
                        {k: (np.array(at(lists, k)) if k in pos else None) for k in pick}, {k: at(infos, k) for k in pick}, hdir)
                chk["ok"] = True
                res["parity_checked"] = chk
            else:
                oc, ok = {}, True
                for k in pick:
                    i, j = mine[k]
                    for q in (i, j):
                        if q not in oc:
                            oc[q] = IcpOracle(frs[q][0], frs[q][1], 0.03)
                    ok = ok and int(oc[j].count_inliers(oc[i], Ts[k], 0.03)) == int(cnts[k])
                res["parity_checked"] = {"pre_check_counts_exact_on_pairs": len(pick), "against": "oracle/icp_oracle.cpp (oracle/_ref did not travel)", "ok": bool(ok)}
        except AssertionError as ex:
            res["parity_checked"] = {"ok": False, "mismatch": str(ex)[:400]}
        except Exception as ex:
            res["parity_checked"] = {"ok": None, "note": "checker failed to run: %s" % ex}
    for c in clouds:
        c.close()
    return res


def fopt_section(device):
    """SURVEY.md 8f-2 figure: Hessian assembly of the reference's FragmentOptimizer (SLAC and rigid modes) for 4 fragments
    of ~250 k points / 6 pairs with exact correspondence lists, GPU (er_fopt_assemble_*) vs the sequential oracle."""
    import numpy as np
    from elasticreconstruction_amd import synth
    from elasticreconstruction_amd.fopt import FragmentOptimizer
    from elasticreconstruction_amd.icp import Cloud, find_correspondence_batch
    num, length = 4, 3.0
    base = synth.look_at((1.5, 1.5, 1.5), (0, 0, 1)) @ np.linalg.inv(synth.basepose())
    frags, poses = [], []
    for f in range(num):
        P = base @ (synth.perturbation(70 + 10 * f, 4.0, 0.06) if f else np.eye(4))
        x, n = synth.sample_fragment(P, 600000, seed=70 + f)
        ok = ((x > 1e-3) & (x < length - 1e-3)).all(1)
        frags.append((x[ok].astype(np.float32), n[ok].astype(np.float32)))
        poses.append(P)
    clouds = [Cloud(x, n, 0.03, device) for x, n in frags]
    ij = [(i, j) for i in range(num) for j in range(i + 1, num)]
    lists, _ = find_correspondence_batch([clouds[j] for i, j in ij], [clouds[i] for i, j in ij],
                                         [np.linalg.inv(poses[i]) @ poses[j] for i, j in ij], 0.015, 0.866)
    pairs = [(i, j, l) for (i, j), l in zip(ij, lists)]
    ncorr = int(sum(l.shape[0] for l in lists))
    g = FragmentOptimizer(num, 8, length, device)
    for f, (x, n) in enumerate(frags):
        g.SetCloud(f, x, n)
        g.UpdatePose(f, poses[f].astype(np.float32))
    groups = g.SetCorrespondences(pairs)
    Rt = np.stack([P[:3, :3].T.reshape(9) for P in poses])
    g.AssembleSLAC(Rt)
    g.AssembleRigid()
    g.AssembleNonrigid(1.0)
    Jb0, _ = g.FactorSLAC(Rt, 4.0)
    g.Solve(Jb0)
    tf = []
    for _ in range(5):                                       # the whole linear step on the device: assemble + base terms + Cholesky + solve
        t0 = time.perf_counter()
        Jb0, _ = g.FactorSLAC(Rt, 4.0)
        g.Solve(Jb0)
        tf.append(time.perf_counter() - t0)
    ts, tr, tn = [], [], []
    for _ in range(5):
        t0 = time.perf_counter()
        JJ, _, _ = g.AssembleSLAC(Rt)
        ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        g.AssembleRigid()
        tr.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        g.AssembleNonrigid(1.0)
        tn.append(time.perf_counter() - t0)
    res = {"correspondences": ncorr, "pairs": len(pairs), "group_chunks": int(groups), "slac_matrix_dim": int(JJ.shape[0]),
           "slac_assembly_ms": 1e3 * sorted(ts)[2], "rigid_assembly_ms": 1e3 * sorted(tr)[2], "nonrigid_assembly_ms": 1e3 * sorted(tn)[2],
           "slac_correspondences_per_s": ncorr / sorted(ts)[2], "slac_assemble_factor_solve_on_device_ms": 1e3 * sorted(tf)[2],
           "what": "er_fopt_assemble_slac / _rigid / _nonrigid = Hessian assembly of OptimizeSLAC / OptimizeRigid / OptimizeNonrigid (OptApp.cpp:473-560, 312-375, 159-206), result copied to the host included"}
    try:
        from oracle.pyoracle import FoptOracle
        o = FoptOracle(num, 8, length)
        for f, (x, n) in enumerate(frags):
            o.set_cloud(f, x, n)
            o.update_pose(f, poses[f].astype(np.float32))
        sub = [(i, j, l[:20000]) for i, j, l in pairs]
        o.set_pairs(sub)
        t0 = time.perf_counter()
        o.assemble_slac(Rt)
        nsub = sum(l.shape[0] for _, _, l in sub)
        res["cpu_port_slac_correspondences_per_s"] = nsub / (time.perf_counter() - t0)
        res["cpu_port_note"] = "oracle/fopt_oracle.cpp, 1 thread, %d correspondences (the reference runs the same loop on 8 OpenMP threads)" % nsub
    except Exception as ex:
        res["cpu_port_note"] = "oracle not available: %s" % ex
    return res


def other_configs(device):
    """BASELINE.json configs[3] and configs[4] at one GPU's share, each as a CHILD run of this file (its own process, volume and
    scene; the headline's numbers are final before this starts): short passes, no CPU legs.  A child that fails or overruns leaves
    its error text here and never touches the headline line."""
    import subprocess
    import time
    res = {"what": "python bench.py --config 4 | 5 --min-seconds 0.2 --cpu-sample 100, run after the headline measurement in child "
                   "processes; one GPU: configs[3] forces the frame-split merge (er_tsdf_allreduce on a 1-rank communicator); "
                   "parity_checked = 100 frames sampled across the job (first fragment to last, true frame ids and lattices) against "
                   "the reference's own CIntegrateApp::Execute, bit for bit"}
    env = dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", str(device)), MASTER_PORT="29531")
    if "HIP_VISIBLE_DEVICES" not in os.environ:
        env["LOCAL_RANK"] = "0"
    for cfg in (4, 5):
        t0 = time.time()
        full = os.path.join(tempfile.gettempdir(), "bench_child_%d_%d.json" % (os.getpid(), cfg))
        try:
            p = subprocess.run([sys.executable, BENCH_PY, "--config", str(cfg), "--min-seconds", "0.2", "--cpu-sample", "100",
                                "--no-alone", "--no-streamed", "--other-configs", "0", "--full-json", full], env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=240)
            if p.returncode != 0 or not os.path.exists(full):
                res["configs[%d]" % (cfg - 1)] = {"error": "rc %d: %s" % (p.returncode, p.stderr.decode()[-400:])}
                continue
            with open(full) as fh:
                d = json.load(fh)
            r = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "scaling": d["scaling"],
                 "workload": d["config"]["workload"], "frames": d["config"]["frames_per_gpu"],
                 "volume_units_touched": d["config"].get("volume_units_touched"),
                 "merge_union_units": d["config"].get("merge_union_units"), "merge_impl": d["config"].get("merge_impl"),
                 "roofline_frac": (d.get("roofline") or {}).get("frac"), "parity_checked": d.get("parity_checked"),
                 "cpu_baseline": d.get("cpu_baseline"), "wall_s": time.time() - t0}
            i = d.get("icp")
            if i:
                r["icp"] = {k: i[k] for k in ("pairs_per_s", "pairs_total", "pairs_this_rank", "accepted_this_rank", "rejected_by_pre_check",
                                              "parity_checked", "fragments") if k in i}
            res["configs[%d]" % (cfg - 1)] = r
        except Exception as ex:                                         # timeout, unparsable output
            res["configs[%d]" % (cfg - 1)] = {"error": repr(ex)[:400]}
        finally:
            if os.path.exists(full):
                os.remove(full)
    return res


def boundary_section(sc, depth, device, png_frames=600, ref_frames=100, ref_pairs=4):
    """SURVEY.md 8b / VERDICT round 5 (7): the throughput a pipeline script sees at the drop-in boundary -- process start, file reads, the GPU work, file
    writes -- for the two programs, each beside the reference's OWN program (oracle/_ref/*_ref, compiled in place) on a bounded sample of the same files:
      bin/Integrate            configs[1] (all frames of `sc`) from a raw uint16 stream, pose.log / seg.log / .ctr in, world.pcd out; and its first
                               `png_frames` frames from a --depth_list of 16-bit PNGs (inflated ahead by host threads);
      Integrate_ref            the first `ref_frames` frames of the same raw stream (8 OpenMP threads as hard-coded);
      bin/BuildCorrespondence  the 50-pair / 25-fragment list of configs[2]: cloud_bin_<i>.pcd in, reg_output.log / .info and 50 corres_<i>_<j>.txt out;
      BuildCorrespondence_ref  the first `ref_pairs` pairs of the same list (it still loads all 25 fragments).
    Wall times of subprocess.run (the fastest of three runs for the two programs: process start-up and exit vary by 0.1-0.2 s on these boxes); inputs live in /dev/shm when it has room (page cache either way)."""
    import shutil
    import subprocess
    import numpy as np
    from elasticreconstruction_amd import formats, synth
    res = {}
    base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 8e9 else None
    d = tempfile.mkdtemp(prefix="er_boundary_", dir=base)
    bin_dir = os.path.join(ROOT, "elasticreconstruction_amd", "bin")
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    env = dict(os.environ, ER_ORACLE_QUIET="1", ER_TIMING="1", HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", str(device)))
    stages = {}

    def timed(cmd, cwd, reps=1, tag=None):
        # (ER_TIMING=1: the programs report the wall time of each of their stages on stderr -- kept for the fastest repetition)
        best, rc, err = None, 0, ""
        for _ in range(reps):
            t0 = time.perf_counter()
            r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env, timeout=600)
            dt = time.perf_counter() - t0
            text = r.stderr.decode(errors="replace")
            rc, err = r.returncode, "\n".join(l for l in text.splitlines() if not l.startswith("[timing]"))[-300:]
            if best is None or dt < best:
                best = dt
                if tag:
                    st = [l[len("[timing]"):].rsplit(None, 2) for l in text.splitlines() if l.startswith("[timing] ") and not l.startswith("[timing]   ")]
                    stages[tag] = {k.strip(): float(v) for k, v, _ in st}
                    stages[tag]["outside main() (exec, loader, exit)"] = round(dt * 1e3 - sum(stages[tag].values()), 1)
                    detail = [l[len("[timing]"):].strip() for l in text.splitlines() if l.startswith("[timing]   ")]
                    if detail:
                        stages[tag]["detail"] = detail[:8]
        return best, rc, err

    try:
        n, I = sc["n"], sc["interval"]
        host = synth.to_numpy_u16(depth)

        def write_inputs(tag, m):
            num = m // I
            pose = [formats.FramedTransformation(i, i, i + 1, sc["pose"][i]) for i in range(num)]
            seg = [formats.FramedTransformation(i, i, i + 1, sc["seg"][i]) for i in range(m)]
            formats.save_log(os.path.join(d, "pose_%s.log" % tag), pose + [formats.FramedTransformation(num, num, num + 1, sc["pose"][num - 1])])
            formats.save_log(os.path.join(d, "seg_%s.log" % tag), seg + [formats.FramedTransformation(m + j, m + j, m + j + 1, sc["seg"][m - 1]) for j in range(I)])
            formats.save_ctr(os.path.join(d, "g_%s.ctr" % tag), sc["grids"][:num])
            return ["--pose_traj", "pose_%s.log" % tag, "--seg_traj", "seg_%s.log" % tag, "--ctr", "g_%s.ctr" % tag, "--num", str(num),
                    "--resolution", str(sc["resolution"]), "--length", str(sc["length"]), "--interval", str(I)]
        host.tofile(os.path.join(d, "frames.raw"))
        a_full = write_inputs("full", n)
        dt, rc, err = timed([os.path.join(bin_dir, "Integrate")] + a_full + ["-oni", "frames.raw", "--save_to", "world.pcd", "--max_units", "1024"], d, reps=3, tag="integrate")
        res["integrate"] = {"frames": n, "source": "raw uint16 stream (%.1f GB)" % (host.nbytes / 1e9), "wall_s": dt, "frames_per_s": n / dt, "rc": rc,
                            "stages_ms": stages.get("integrate")}
        if rc:
            res["integrate"]["stderr"] = err
        try:
            from PIL import Image
            m = min(n, max(I, png_frames // I * I))
            t0 = time.perf_counter()
            with open(os.path.join(d, "list.txt"), "w") as f:
                for i in range(m):
                    Image.fromarray(host[i].reshape(480, 640)).save(os.path.join(d, "f%05d.png" % i), compress_level=1)
                    f.write("f%05d.png\n" % i)
            t_png = time.perf_counter() - t0
            a_png = write_inputs("png", m)
            dt, rc, err = timed([os.path.join(bin_dir, "Integrate")] + a_png + ["--depth_list", "list.txt", "--save_to", "world_png.pcd", "--max_units", "1024"], d, reps=3, tag="integrate_png")
            res["integrate_png"] = {"frames": m, "wall_s": dt, "frames_per_s": m / dt, "rc": rc, "decode_threads": "default (hardware threads / 8, 8..32)", "png_written_in_s": t_png,
                                    "stages_ms": stages.get("integrate_png"), "host_hardware_threads": os.cpu_count()}
            dt1, rc1, _ = timed([os.path.join(bin_dir, "Integrate")] + a_png + ["--depth_list", "list.txt", "--save_to", "world_png.pcd", "--max_units", "1024",
                                                                                 "--decode_threads", "1"], d)
            res["integrate_png"]["one_decode_thread_frames_per_s"] = m / dt1
            dt8, rc8, _ = timed([os.path.join(bin_dir, "Integrate")] + a_png + ["--depth_list", "list.txt", "--save_to", "world_png.pcd", "--max_units", "1024",
                                                                                 "--decode_threads", "8"], d)
            res["integrate_png"]["eight_decode_threads_frames_per_s"] = m / dt8
        except Exception as ex:
            res["integrate_png"] = {"error": repr(ex)[:200]}
        ref_bin = os.path.join(ref_dir, "Integrate_ref")
        if os.path.exists(ref_bin):
            m = min(n, max(I, ref_frames // I * I))
            host[:m].tofile(os.path.join(d, "frames_ref.raw"))
            a_ref = write_inputs("ref", m)
            dt, rc, err = timed([ref_bin] + a_ref + ["-oni", "frames_ref.raw", "--save_to", "world_ref.pcd"], d)
            res["integrate_reference_program"] = {"frames": m, "wall_s": dt, "frames_per_s": m / dt, "rc": rc, "threads": 8}
        # ---- BuildCorrespondence ----
        frs = synth.fragment_set(25, 250000, device="cuda:%d" % device)
        for i, (x, nn, _) in enumerate(frs):
            formats.save_pcd_xyzn(os.path.join(d, "cloud_bin_%d.pcd" % i), x, nn, binary=True)
        pairs = synth.config2_pair_list(frs, 50)
        log = [formats.FramedTransformation(a, b, len(frs), T) for a, b, T in pairs]
        formats.save_log(os.path.join(d, "init.log"), log)
        formats.save_log(os.path.join(d, "init_ref.log"), log[:ref_pairs])
        bc = ["--registration", "--reg_dist", "0.03", "--output_information"]
        dt, rc, err = timed([os.path.join(bin_dir, "BuildCorrespondence"), "--reg_traj", os.path.join(d, "init.log")] + bc, d, reps=3, tag="bc")
        out_bytes = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d) if f.startswith("corres_"))
        res["build_correspondence"] = {"pairs": len(pairs), "fragments": len(frs), "wall_s": dt, "pairs_per_s": len(pairs) / dt, "rc": rc,
                                       "pcd_bytes_read": sum(os.path.getsize(os.path.join(d, "cloud_bin_%d.pcd" % i)) for i in range(len(frs))),
                                       "corres_txt_bytes_written": out_bytes, "stages_ms": stages.get("bc")}
        if rc:
            res["build_correspondence"]["stderr"] = err
        ref_bin = os.path.join(ref_dir, "BuildCorrespondence_ref")
        if os.path.exists(ref_bin):
            dt, rc, err = timed([ref_bin, "--reg_traj", os.path.join(d, "init_ref.log")] + bc, d)
            res["build_correspondence_reference_program"] = {"pairs": ref_pairs, "wall_s": dt, "pairs_per_s": ref_pairs / dt, "rc": rc, "threads": 8,
                                                             "note": "loads all %d fragments, then %d pairs" % (len(frs), ref_pairs)}
    except Exception as ex:
        res["error"] = repr(ex)[:300]
    finally:
        shutil.rmtree(d, ignore_errors=True)
    g = lambda *k: _dig(res, *k)
    res["compact"] = {"integrate_fps": g("integrate", "frames_per_s"), "integrate_png_fps": g("integrate_png", "frames_per_s"),
                      "integrate_ref_fps": g("integrate_reference_program", "frames_per_s"), "bc_pairs_per_s": g("build_correspondence", "pairs_per_s"),
                      "bc_ref_pairs_per_s": g("build_correspondence_reference_program", "pairs_per_s")}
    return res


def _dig(d, *path):
    for p in path:
        if not isinstance(d, dict) or d.get(p) is None:
            return None
        d = d[p]
    return d
