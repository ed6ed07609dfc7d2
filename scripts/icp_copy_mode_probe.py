#!/usr/bin/env python3
"""What puts the per-list D2H copies of er_find_correspondence_batch into their slow mode (4.4 against 1.6-1.9 ms per 50-pair list)?  The uniform list with
ER_ICP_DIRECT_LISTS=0 in a fresh process, after one optional step: usage: python scripts/icp_copy_mode_probe.py [none|volume|volume_closed|integrate|torch|volumes25|read_units|pageable|kinfu1]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["ER_ICP_DIRECT_LISTS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
what = sys.argv[1] if len(sys.argv) > 1 else "none"
from elasticreconstruction_amd import synth
cache = "/tmp/er_icp_ab_cache.npz"
if os.path.exists(cache):
    z = np.load(cache)
    frs = [(z["ux%d" % i], z["un%d" % i], z["uF%d" % i]) for i in range(25)]
else:
    frs = [(np.ascontiguousarray(x), np.ascontiguousarray(n), F) for x, n, F in synth.fragment_set(25, 250000, device="cuda:0")]
    np.savez(cache, **{("u%s%d" % (t, i)): a for i, fr in enumerate(frs) for t, a in zip("xnF", fr)})
keep = []
if what in ("volume", "volume_closed", "integrate"):
    from elasticreconstruction_amd.tsdf import TSDFVolume
    v = TSDFVolume(max_units=640, device=0)
    if what == "integrate":
        sc = synth.make_scenario(50, interval=50, warp=True, device="cuda:0")
        v.IntegrateFrames(synth.to_numpy_u16(sc["depth"]), sc["traj"], synth.warp_arrays(sc))
        v.synchronize()
    if what == "volume_closed":
        v.close()
    else:
        keep.append(v)
if what == "volumes25":
    from elasticreconstruction_amd.tsdf import TSDFVolume
    for _ in range(25):
        TSDFVolume(max_units=1024, device=0).close()
if what == "read_units":
    from elasticreconstruction_amd.tsdf import TSDFVolume
    v = TSDFVolume(max_units=640, device=0)
    sc = synth.make_scenario(50, interval=50, warp=True, device="cuda:0")
    v.IntegrateFrames(synth.to_numpy_u16(sc["depth"]), sc["traj"], synth.warp_arrays(sc))
    for k in v.unit_keys():
        v.read_unit(int(k))
    v.extract_surface()
    v.close()
if what == "pageable":
    import torch
    for _ in range(200):
        torch.from_numpy(np.zeros((64, 64, 64), np.float32)).to("cuda:0")
    big = torch.zeros((512, 512, 512), device="cuda:0")
    h = big[:64].cpu()
    del big
    torch.cuda.synchronize()
if what == "kinfu1":
    synth.kinfu_fragment(0, 50, 250000)
if what in ("streams100", "streams_kept", "hostmalloc100", "malloc25"):
    import ctypes
    import torch
    torch.cuda.init()
    hip = ctypes.CDLL("libamdhip64.so")
    vp = ctypes.c_void_p
    if what == "streams100":                 # what 25 volumes do to the runtime's streams: 4 created, 4 destroyed, 25 times
        for _ in range(25):
            ss = [vp() for _ in range(4)]
            for q in ss:
                assert hip.hipStreamCreateWithFlags(ctypes.byref(q), 1) == 0
            for q in ss:
                assert hip.hipStreamDestroy(q) == 0
    if what == "streams_kept":               # 12 streams created and kept alive
        for _ in range(12):
            q = vp()
            assert hip.hipStreamCreateWithFlags(ctypes.byref(q), 1) == 0
            keep.append(q)
    if what == "hostmalloc100":
        for _ in range(100):
            q = vp()
            assert hip.hipHostMalloc(ctypes.byref(q), ctypes.c_size_t(64 << 20), 0) == 0
            assert hip.hipHostFree(q) == 0
    if what == "malloc25":
        for _ in range(25):
            q = vp()
            assert hip.hipMalloc(ctypes.byref(q), ctypes.c_size_t(2 << 30)) == 0
            assert hip.hipFree(q) == 0
if what == "torch":
    import torch
    keep.append(torch.zeros(1 << 20, device="cuda:0"))
    torch.cuda.synchronize()
from elasticreconstruction_amd.icp import Cloud, count_inliers_batch, find_correspondence_batch, icp_align_batch
cl = [Cloud(x, n, 0.03, 0) for x, n, _ in frs]
pairs = synth.config2_pair_list(frs, 50)
srcs, tgts = [cl[b] for _, b, _ in pairs], [cl[a] for a, _, _ in pairs]
def churn():
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    for _ in range(25):
        q = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(q), ctypes.c_size_t(2 << 30)) == 0
        assert hip.hipFree(q) == 0


ph = []
for r in range(7 if what != "warm_then_malloc25" else 12):
    if what == "warm_then_malloc25" and r == 5:                # workspaces, arena and streams exist already: is it where the buffers land, or the process?
        churn()
    t0 = time.perf_counter()
    count_inliers_batch(srcs, tgts, [T for _, _, T in pairs], 0.03)
    t1 = time.perf_counter()
    fins, iters, conv, _ = icp_align_batch(srcs, tgts, [T.astype(np.float32) for _, _, T in pairs], 0.03, 20, 1e-6, 0)
    t2 = time.perf_counter()
    find_correspondence_batch(srcs, tgts, [F.astype(np.float64) for F in fins], 0.015, 0.8660, True, copy=False)
    t3 = time.perf_counter()
    ph.append((t1 - t0, t2 - t1, t3 - t2))
if what == "warm_then_malloc25":
    print("fc per pass (churn before pass 5): %s" % np.round(np.array(ph)[:, 2] * 1e3, 2))
ph = np.array(ph[2:] if what != "warm_then_malloc25" else ph[7:]) * 1e3
print("%-14s pre/icp/fc median %s" % (what, np.round(np.median(ph, 0), 2)), flush=True)
