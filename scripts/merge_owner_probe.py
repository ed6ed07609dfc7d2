#!/usr/bin/env python3
"""The 8-rank frame-split merge of BASELINE.json configs[3] (10 000 frames of the drifting path through the 6 m room, 8 contiguous blocks) on ONE GPU through
the loopback communicator -- everything of er_tsdf_allreduce except the wire -- once per protocol:
  owner / distributed   round 6: band records to the unit owners, rank-ordered sums, result left distributed (what bench.py --config 4 --gpus 8 times)
  owner / root 0        ... then gathered on rank 0
  ring / root 0         round 5: whole [sdf*w | w] planes through one sum + raw units point to point
For each: bytes moved (er_comm_merge_stats[_owner], summed over the ranks), wall time with all eight ranks sharing the GPU, and the merged volume against ONE
volume that integrated all 10 000 frames: key set, weights (exact), sdf (<= 1e-5).  usage: python scripts/merge_owner_probe.py [ranks] (one MI355X, ~1.5 min)"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from elasticreconstruction_amd import parallel, synth
from elasticreconstruction_amd.tsdf import TSDFVolume

dev = torch.device("cuda", 0)
N, G = 10000, int(sys.argv[1]) if len(sys.argv) > 1 else 8
per = -(-N // (G * 50)) * 50


def integrate_all(targets_of_rank):
    for r in range(G):
        sc = synth.make_scenario(per, interval=50, warp=True, frame_offset=r * per, total_frames=N, revolutions=N / 3000.0, radius_drift=1.5, room=(-1.5, 4.5), device=dev)
        w = synth.warp_arrays(sc)
        torch.cuda.synchronize()
        px = sc["depth"].shape[1]
        for lo in range(0, per, 200):
            hi = lo + 200
            gi = w["grid_index"][lo:hi]
            g0, g1 = int(gi.min()), int(gi.max()) + 1
            ws = dict(ctr=w["ctr"][g0:g1], resolution=w["resolution"], length=w["length"], grid_index=gi - g0, seg=w["seg"][lo:hi], madj=w["madj"][lo:hi])
            for tgt in targets_of_rank(r):
                tgt.IntegrateFrames(None, sc["traj"][lo:hi], ws, device_ptr=sc["depth"].data_ptr() + lo * px * 2)
        for tgt in targets_of_rank(r):
            tgt.synchronize()
        del sc


full = TSDFVolume(max_units=2048, device=0)
integrate_all(lambda r: [full])
keys = [int(k) for k in full.unit_keys()]
sample = keys[:: max(1, len(keys) // 150)]
ref = {k: full.read_unit(k) for k in sample}
obs = one = wmax = 0
for k in sample:                                            # what a band record could still shed: observed voxels whose sdf is exactly 1 (free space in front of a surface)
    s_, w_ = ref[k]
    obs += int((w_ != 0).sum())
    one += int(((w_ != 0) & (s_ == 1.0)).sum())
    wmax = max(wmax, float(w_.max()))
out = {"ranks": G, "frames": N, "union_units": len(keys), "sum_weight": full.sum_weight(),
       "sampled_units": len(sample), "observed_fraction": obs / (len(sample) * 64.0 ** 3), "observed_with_sdf_exactly_1": one / max(obs, 1), "max_weight": wmax}
print(json.dumps(out), flush=True)
for impl, root in (("owner", parallel.MERGE_DISTRIBUTED), ("owner", 0), ("ring", 0)):
    os.environ["ER_MERGE_IMPL"] = impl
    vols = [TSDFVolume(max_units=2048, device=0) for _ in range(G)]
    integrate_all(lambda r: [vols[r]])
    comms = parallel.LoopbackComms(G)
    times = []
    for rep in range(2):                                    # the second merge meets warm plane / record buffers: that one is timed
        if rep:
            for v in vols:
                v.reset()
            integrate_all(lambda r: [vols[r]])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nu = comms.allreduce(vols, root=root)
        times.append(time.perf_counter() - t0)
    times = times[1:]
    st = [comms.merge_stats(r) for r in range(G)]
    have = [set(int(k) for k in v.unit_keys()) for v in vols]
    where = {k: [r for r in range(G) if k in have[r]] for k in keys}
    if root == parallel.MERGE_DISTRIBUTED:
        ok_keys = all(len(where[k]) == 1 for k in keys) and set().union(*have) == set(keys)
    else:
        ok_keys = have[root] == set(keys)
    worst, wdiff = 0.0, 0
    for k in sample:
        r = where[k][0] if root == parallel.MERGE_DISTRIBUTED else root
        sm, wm = vols[r].read_unit(k)
        wdiff += int((ref[k][1] != wm).sum())
        worst = max(worst, float(np.abs(ref[k][0] - sm).max()))
    sw = sum(v.sum_weight() for v in vols) if root == parallel.MERGE_DISTRIBUTED else vols[root].sum_weight()
    e = {"union": int(nu), "multi_toucher_units": st[0]["multi_toucher_units"], "single_toucher_units": st[0]["single_toucher_units"],
         "bytes_sent_all_ranks": sum(s["bytes_sent"] for s in st), "bytes_received_max_rank": max(s["bytes_received"] for s in st),
         "bytes_reduced_per_rank": st[0]["bytes_reduced"], "merge_ms_all_ranks_on_one_gpu": round(1e3 * min(times), 2),
         "keys_ok": bool(ok_keys), "sum_weight_equal": bool(sw == out["sum_weight"]), "weight_mismatches_in_sampled_units": wdiff, "max_abs_sdf_diff_in_sampled_units": worst,
         "units_per_rank_after": [len(h) for h in have]}
    if impl == "owner":
        e.update({"to_owners_bytes": sum(s["to_owners_bytes_sent"] for s in st), "to_root_bytes": sum(s["to_root_bytes_sent"] for s in st),
                  "to_owners_bytes_received_max_rank": max(s["to_owners_bytes_received"] for s in st),
                  "ring_equivalent_bytes": st[0]["ring_equivalent_bytes"], "units_summed_per_rank": [s["units_summed_here"] for s in st]})
    out["%s/%s" % (impl, "distributed" if root == parallel.MERGE_DISTRIBUTED else "root%d" % root)] = e
    print(impl, root, json.dumps(e), flush=True)
    comms.close()
    for v in vols:
        v.close()
full.close()
print(json.dumps(out))
