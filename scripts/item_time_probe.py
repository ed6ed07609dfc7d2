#!/usr/bin/env python3
"""Probe (patched library variant 'dbg', scripts/gpu_r3C.sh): per-workgroup start / end wall-clock stamps and the longest item of the LAST
k_integrate launch of a 150-frame step of the bench scene: how much of the kernel is tail?"""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from elasticreconstruction_amd import synth, _ffi
from elasticreconstruction_amd.tsdf import TSDFVolume

n, I = 600, 50
dev = torch.device("cuda", 0)
sc = synth.make_scenario(n, interval=I, warp=True, frame_offset=0, total_frames=3000, revolutions=1.0, device=dev)
warp_all = synth.warp_arrays(sc)
depth = sc["depth"]
px = depth.shape[1]
vol = TSDFVolume(max_units=512, device=0)
lib = _ffi.lib()
buf = (C.c_ulonglong * (2048 * 6))()
for rep in range(2):
    vol.reset()
    for lo in range(0, n, I):
        hi = lo + I
        gi = warp_all["grid_index"][lo:hi]
        g0, g1 = int(gi.min()), int(gi.max()) + 1
        w = dict(ctr=warp_all["ctr"][g0:g1], resolution=warp_all["resolution"], length=warp_all["length"], grid_index=gi - g0,
                 seg=warp_all["seg"][lo:hi], madj=warp_all["madj"][lo:hi])
        vol.IntegrateFrames(None, sc["traj"][lo:hi], w, device_ptr=depth.data_ptr() + lo * px * 2)
        vol.synchronize()                       # one launch at a time: the kernel alone
        if rep == 1 and lo >= n - 3 * I:
            assert lib.er_debug_read(buf) == 0
            a = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 6)[:512].astype(np.int64)
            t0 = a[:, 0].min()
            start, end, longest = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, a[:, 2] / 100.0     # us (100 MHz clock)
            print("launch at frame %4d: %d items; workgroups start %.1f..%.1f us, end min %.1f  mean %.1f  p90 %.1f  max %.1f us; "
                  "items per workgroup %.1f (min %d max %d)" % (lo, a[0, 5], start.min(), start.max(), end.min(), end.mean(),
                                                             np.percentile(end, 90), end.max(), a[:, 4].mean(), a[:, 4].min(), a[:, 4].max()))
            o = np.argsort(-longest)[:8]
            print("   longest items (us, item index = unit rank * 128 + sub-block):", [(round(float(longest[k]), 1), int(a[k, 3])) for k in o])
            late = np.argsort(-end)[:8]
            print("   last workgroups: end (us), their longest item (us):", [(round(float(end[k]), 1), round(float(longest[k]), 1)) for k in late])
            print("   sum of busy time / (512 x kernel duration) = %.2f" % ((end - start).sum() / (512 * end.max())))
vol.close()
