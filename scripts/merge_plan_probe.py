#!/usr/bin/env python3
"""What the frame-split merge of BASELINE.json configs[1] (weak: every rank one 3000-frame revolution) and configs[3] (strong: 10 000 frames of the
drifting path cut into G contiguous blocks) has to move at G = 1, 2, 4, 8 -- measured on ONE GPU by integrating every rank's block into a volume of its own
and reading the unit key sets: union, units two or more ranks touched (they go through the sum reduction), units one rank touched (they travel point to
point to rank 0, or stay), bytes against the dense protocol of rounds 2-4 -- together with the single-GPU terms a prediction of the N-GPU line needs:
each rank's compute time for its block, and the export / import kernels' rate.  DESIGN.md section 6 turns this into the predicted frames/s.
usage: python scripts/merge_plan_probe.py  (one MI355X; ~1 min)"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from elasticreconstruction_amd import synth
from elasticreconstruction_amd.tsdf import TSDFVolume

dev = torch.device("cuda", 0)
UNIT_BYTES = 2 * 64 ** 3 * 4
out = {}


def run_block(sc, max_units, step):
    """integrate a scenario's frames in steps of `step`; returns (keys, seconds of the second pass)"""
    depth, px = sc["depth"], sc["depth"].shape[1]
    warp = synth.warp_arrays(sc)
    torch.cuda.synchronize()                                   # (the frames are rendered on torch's stream)
    vol = TSDFVolume(max_units=max_units, device=0)
    dt = None
    for rep in range(2):
        vol.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for lo in range(0, sc["n"], step):
            hi = min(lo + step, sc["n"])
            gi = warp["grid_index"][lo:hi]
            g0, g1 = int(gi.min()), int(gi.max()) + 1
            w = dict(ctr=warp["ctr"][g0:g1], resolution=warp["resolution"], length=warp["length"], grid_index=gi - g0, seg=warp["seg"][lo:hi], madj=warp["madj"][lo:hi])
            vol.IntegrateFrames(None, sc["traj"][lo:hi], w, device_ptr=depth.data_ptr() + lo * px * 2)
        vol.synchronize()
        dt = time.perf_counter() - t0
    keys = vol.unit_keys().copy()
    return vol, keys, dt


def plan(keysets, root=0):
    allk = np.concatenate(keysets)
    u, cnt = np.unique(allk, return_counts=True)
    multi = u[cnt >= 2]
    single = u[cnt == 1]
    own_root = np.intersect1d(single, keysets[root])
    travel = len(single) - len(own_root)
    return {"union": int(len(u)), "multi_toucher": int(len(multi)), "single_toucher": int(len(single)), "single_on_root": int(len(own_root)), "single_travelling": int(travel),
            "toucher_histogram": {int(k): int(v) for k, v in zip(*np.unique(cnt, return_counts=True))},
            "bytes_reduced": int(len(multi)) * UNIT_BYTES, "bytes_point_to_point": int(travel) * UNIT_BYTES, "bytes_dense_protocol": int(len(u)) * UNIT_BYTES,
            "units_per_rank": [int(len(k)) for k in keysets]}


# ---- configs[3]: 10 000 frames, strong scaling ------------------------------------------------------------------------------------------------------
N = 10000
res3 = {}
for G in (1, 2, 4, 8):
    per = -(-N // (G * 50)) * 50
    keysets, secs = [], []
    for r in range(G):
        n = min(per, N - r * per)
        sc = synth.make_scenario(n, interval=50, warp=True, frame_offset=r * per, total_frames=N, revolutions=N / 3000.0, radius_drift=1.5, room=(-1.5, 4.5), device=dev)
        vol, keys, dt = run_block(sc, 4096, 200)
        vol.close()
        del sc
        keysets.append(keys)
        secs.append(dt)
    p = plan(keysets)
    p.update({"frames_per_rank": per, "compute_ms_per_rank": [round(1e3 * s, 2) for s in secs], "compute_ms_slowest": round(1e3 * max(secs), 2)})
    res3[G] = p
    print("configs[3] G=%d: %s" % (G, json.dumps(p)), flush=True)
out["configs[3]"] = res3

# ---- the REAL merge of configs[3] with 8 ranks on this one GPU (loopback communicator: the protocol, the device volumes, the export / import kernels and the
# plane buffers of er_tsdf_allreduce; a summing kernel and device-to-device copies where RCCL would be) -- everything of the 8-rank merge except the wire ----
try:
    from elasticreconstruction_amd import parallel
    G = 8
    per = -(-N // (G * 50)) * 50
    vols, full = [], TSDFVolume(max_units=2048, device=0)
    for r in range(G):
        sc = synth.make_scenario(per, interval=50, warp=True, frame_offset=r * per, total_frames=N, revolutions=N / 3000.0, radius_drift=1.5, room=(-1.5, 4.5), device=dev)
        w = synth.warp_arrays(sc)
        torch.cuda.synchronize()
        v = TSDFVolume(max_units=2048 if r == 0 else 1024, device=0)
        px = sc["depth"].shape[1]
        for lo in range(0, per, 200):
            hi = lo + 200
            gi = w["grid_index"][lo:hi]
            g0, g1 = int(gi.min()), int(gi.max()) + 1
            ws = dict(ctr=w["ctr"][g0:g1], resolution=w["resolution"], length=w["length"], grid_index=gi - g0, seg=w["seg"][lo:hi], madj=w["madj"][lo:hi])
            for tgt in (v, full):
                tgt.IntegrateFrames(None, sc["traj"][lo:hi], ws, device_ptr=sc["depth"].data_ptr() + lo * px * 2)
        v.synchronize(); full.synchronize()
        vols.append(v)
        del sc
    comms = parallel.LoopbackComms(G)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    nu = comms.allreduce(vols, root=0)
    t_merge = time.perf_counter() - t0
    st = comms.merge_stats(0)
    worst, wdiff = 0.0, 0
    keys = full.unit_keys()
    same_keys = bool(np.array_equal(keys, vols[0].unit_keys()))
    for k in keys[:: max(1, len(keys) // 120)]:               # every ninth unit or so: 120 units x 2 MiB x 2 read back
        sf, wf = full.read_unit(k)
        sm, wm = vols[0].read_unit(k)
        wdiff += int((wf != wm).sum())
        worst = max(worst, float(np.abs(sf - sm).max()))
    out["loopback_merge_8_ranks"] = {"union": int(nu), "merge_ms_without_a_wire": round(1e3 * t_merge, 2), "stats_root": st, "keys_equal_single_volume": same_keys,
                                     "weight_mismatches_in_sampled_units": wdiff, "max_abs_sdf_diff_in_sampled_units": worst}
    print("loopback merge of configs[3], 8 ranks on one GPU:", json.dumps(out["loopback_merge_8_ranks"]), flush=True)
    comms.close()
    for v in vols + [full]:
        v.close()
except Exception as ex:
    print("loopback merge failed:", repr(ex), flush=True)

# ---- configs[1]: every rank one 3000-frame revolution (bench.py --gpus G: frame_offset = rank * 3000 of a G x 3000-frame trajectory) ------------------
res1 = {}
for G in (1, 2, 8):
    keysets, secs = [], []
    for r in range(G if G < 8 else 3):                      # (the revolutions are identical up to the trajectory's phase: three ranks of eight are enough to see it)
        sc = synth.make_scenario(3000, interval=50, warp=True, frame_offset=r * 3000, total_frames=G * 3000, revolutions=float(G), device=dev)
        vol, keys, dt = run_block(sc, 1024, 150)
        keysets.append(keys)
        secs.append(dt)
        if r == 0 and G == 1:
            # the export / import kernels on this volume's units: what the root pays before and after the reduction
            nk = len(keys)
            buf = torch.empty((nk, 2, 64 ** 3), dtype=torch.float32, device=dev)
            tt = {}
            for name, fn in (("export_weighted", vol.export_weighted), ("import_weighted", vol.import_weighted), ("export_raw", vol.export_raw), ("import_raw", vol.import_raw)):
                fn(keys, buf.data_ptr()); vol.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    fn(keys, buf.data_ptr())
                vol.synchronize()
                tt[name] = (time.perf_counter() - t0) / 5
            out["export_import"] = {"units": int(nk), "ms": {k: round(1e3 * v, 3) for k, v in tt.items()},
                                    "GB_per_s_of_planes": {k: round(nk * UNIT_BYTES / v / 1e9, 1) for k, v in tt.items()}}
            print("export / import:", json.dumps(out["export_import"]), flush=True)
            del buf
        vol.close()
        del sc
    p = plan(keysets)
    p.update({"ranks_integrated": len(keysets), "compute_ms_per_rank": [round(1e3 * s, 2) for s in secs]})
    res1[G] = p
    print("configs[1] G=%d: %s" % (G, json.dumps(p)), flush=True)
out["configs[1]"] = res1
print(json.dumps(out))
