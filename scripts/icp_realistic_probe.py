#!/usr/bin/env python3
"""Why does bench.py's kinfu-like leg (icp.realistic) take 9.1 ms per list when the same list takes 6.3 ms in scripts/icp_ab.py?  The kinfu list's three
phases, timed after each of the legs bench_extras.icp_section runs before it: usage: python scripts/icp_realistic_probe.py"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from elasticreconstruction_amd import _ffi, synth
from elasticreconstruction_amd.icp import Cloud, count_inliers, count_inliers_batch, find_correspondence, find_correspondence_batch, icp_align, icp_align_batch, registration_batch
n_pairs, n_frag = 50, 25
kfr = []
for i in range(n_frag):
    x, n, F, st = synth.kinfu_fragment(i, 2 * n_frag, 250000, noise_mm=2.0 if i % 2 else 0.0)
    ok = ~np.isnan(n).any(axis=1)
    kfr.append((np.ascontiguousarray(x[ok]), np.ascontiguousarray(n[ok]), F))
kpairs = synth.chain_pair_list(kfr, n_pairs, 2.0, 0.02, 700)


def run_list(plist, cl, tag):
    srcs, tgts = [cl[b] for _, b, _ in plist], [cl[a] for a, _, _ in plist]
    ph = []
    for r in range(7):
        t0 = time.perf_counter()
        count_inliers_batch(srcs, tgts, [T for _, _, T in plist], 0.03)
        t1 = time.perf_counter()
        fins, iters, conv, _ = icp_align_batch(srcs, tgts, [T.astype(np.float32) for _, _, T in plist], 0.03, 20, 1e-6, 0)
        t2 = time.perf_counter()
        lists, infos = find_correspondence_batch(srcs, tgts, [F.astype(np.float64) for F in fins], 0.015, 0.8660, True, copy=False)
        t3 = time.perf_counter()
        ph.append((t1 - t0, t2 - t1, t3 - t2))
    ph = np.array(ph[2:]) * 1e3
    print("%-44s pre/icp/fc median %s  passes %s" % (tag, np.round(np.median(ph, 0), 2), np.round(ph.sum(1), 2)), flush=True)
    return fins


kcl = [Cloud(x, n, 0.03, 0) for x, n, _ in kfr]
run_list(kpairs, kcl, "kinfu list, fresh process")
frs = synth.fragment_set(n_frag, 250000, device="cuda:0")
ucl = [Cloud(x, n, 0.03, 0) for x, n, _ in frs]
upairs = synth.config2_pair_list([(x, n, F) for x, n, F in frs], n_pairs)
run_list(upairs, ucl, "uniform list")
run_list(kpairs, kcl, "kinfu list after the uniform list")
from concurrent.futures import ThreadPoolExecutor


def run_pair(a, b, T):
    tgt, src = ucl[a], ucl[b]
    count_inliers(src, tgt, T, 0.03)
    fin, iters, conv, _ = icp_align(src, tgt, T.astype(np.float32), 0.03, 20, 1e-6, 0)
    find_correspondence(src, tgt, fin.astype(np.float64), 0.015, 0.8660, True)


with ThreadPoolExecutor(8) as ex:
    list(ex.map(lambda p: run_pair(*p), upairs))
run_list(kpairs, kcl, "kinfu list after 8 host threads x single pairs")
run_list(kpairs, kcl, "kinfu list again")
for _ in range(3):
    registration_batch([ucl[b] for _, b, _ in upairs], [ucl[a] for a, _, _ in upairs], [T for _, _, T in upairs], 0.03, 40000, 0.25, 20, 1e-6, 0, 0.015, 0.8660, want_info=True, copy=False)
run_list(kpairs, kcl, "kinfu list after er_registration_batch")
run_list(upairs, ucl, "uniform list again")
_ffi.lib().er_icp_release_workspaces()
run_list(kpairs, kcl, "kinfu list after er_icp_release_workspaces")
