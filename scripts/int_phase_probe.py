#!/usr/bin/env python3
"""Probe (library variant built with -DER_INT_PROBE=1, selected with ER_HIP_LIB): where the waves of k_integrate spend their life on the bench scene -- claim + item
barriers, loads + culling preamble, frame loop, store -- summed over all waves, kernel alone (one launch at a time) and inside the three-stream pipeline."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from elasticreconstruction_amd import synth, _ffi
from elasticreconstruction_amd.tsdf import TSDFVolume

n, I = 3000, 50
dev = torch.device("cuda", 0)
sc = synth.make_scenario(n, interval=I, warp=True, frame_offset=0, total_frames=3000, revolutions=1.0, device=dev)
warp_all = synth.warp_arrays(sc)
depth = sc["depth"]
px = depth.shape[1]
vol = TSDFVolume(max_units=512, device=0)
lib = _ffi.lib()
buf = (C.c_ulonglong * 8)()


def run(step, sync_each):
    vol.reset()
    for lo in range(0, n, step):
        hi = min(lo + step, n)
        gi = warp_all["grid_index"][lo:hi]
        g0, g1 = int(gi.min()), int(gi.max()) + 1
        w = dict(ctr=warp_all["ctr"][g0:g1], resolution=warp_all["resolution"], length=warp_all["length"], grid_index=gi - g0,
                 seg=warp_all["seg"][lo:hi], madj=warp_all["madj"][lo:hi])
        vol.IntegrateFrames(None, sc["traj"][lo:hi], w, device_ptr=depth.data_ptr() + lo * px * 2)
        if sync_each:
            vol.synchronize()
    vol.synchronize()


for label, step, sync_each in (("alone (50-frame launches, one at a time)", 50, True), ("pipeline (150-frame calls, three streams)", 150, False)):
    run(step, sync_each)                         # warm
    assert lib.er_debug_read(buf) == 0
    run(step, sync_each)
    assert lib.er_debug_read(buf) == 0
    life, park, pre, loop, store, waves, items, frames = [int(x) for x in buf]
    print("%s: %d wave lives, %.1f items per wave, %.1f frames per (wave, item) after the culling" % (label, waves, items / max(waves, 1), frames / max(items, 1)))
    print("   share of the wave life: claim + item barriers %.3f | loads + culling preamble %.3f | frame loop %.3f | store %.3f | rest %.3f" % (
        park / life, pre / life, loop / life, store / life, 1.0 - (park + pre + loop + store) / life))
    print("   ticks per frame visit in the loop: %.0f" % (loop / max(frames, 1)))
vol.close()
