"""Shared helpers of the parity tests."""
import hashlib
import json
import os

import numpy as np

from elasticreconstruction_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def golden():
    with open(os.path.join(GOLD, "tsdf_golden.json")) as f:
        return json.load(f)


def golden_rigid():
    """(poses [6,4,4], depth uint16 [6, 307200]) of tests/golden, depth digest verified."""
    z = np.load(os.path.join(GOLD, "tsdf_inputs.npz"))
    poses = z["rigid_poses"]
    depth = synth.to_numpy_u16(synth.render_depth(poses))
    assert digest(depth) == golden()["rigid_depth"], "synthetic renderer is not bit-reproducible on this host"
    return poses, depth


def golden_warp():
    """Scenario dict (as synth.make_scenario) rebuilt from the committed golden inputs."""
    z = np.load(os.path.join(GOLD, "tsdf_inputs.npz"))
    depth_t = synth.render_depth(z["warp_world"])
    depth = synth.to_numpy_u16(depth_t)
    assert digest(depth) == golden()["warp_depth"], "synthetic renderer is not bit-reproducible on this host"
    return dict(depth=depth_t, traj=z["warp_traj"], pose=z["warp_pose"], seg=z["warp_seg"], grids=z["warp_grids"],
                interval=int(z["warp_meta"][0]), resolution=int(z["warp_meta"][1]), length=float(z["warp_length"]),
                n=z["warp_traj"].shape[0])


def volume_digest(vol):
    """Same digest as tests/golden/make_golden.py::volume_digest for any object with unit_keys/read_unit."""
    keys = vol.unit_keys()
    h = hashlib.sha256()
    wsum = 0.0
    for k in keys:
        sdf, w = vol.read_unit(k)
        h.update(np.int32(k).tobytes())
        h.update(sdf.tobytes())
        h.update(w.tobytes())
        wsum += float(w.astype(np.float64).sum())
    return dict(keys=[int(k) for k in keys], sha256=h.hexdigest(), sum_weight=wsum)


def assert_volumes_identical(a, b, what=""):
    """Unit key sets equal and every sdf_/weight_ array bit-identical."""
    ka, kb = a.unit_keys(), b.unit_keys()
    assert np.array_equal(ka, kb), "%s unit key sets differ: %d vs %d units" % (what, len(ka), len(kb))
    for k in ka:
        sa, wa = a.read_unit(k)
        sb, wb = b.read_unit(k)
        nw = int((wa != wb).sum())
        ns = int((sa.view(np.uint32) != sb.view(np.uint32)).sum())
        assert nw == 0 and ns == 0, "%s unit %d: %d weight_ and %d sdf_ voxels differ (max |dsdf| %.3g)" % (
            what, k, nw, ns, float(np.abs(sa - sb).max()))
    return len(ka)


def oracle_run(ora, sc, depth, warp=None):
    """Frame-by-frame reference flow on the oracle: [Reproject] -> ScaleDepth -> Integrate (IntegrateApp.cpp:217-225)."""
    for f in range(depth.shape[0]):
        d = depth[f]
        if warp is not None:
            d = ora.Reproject(d, warp["ctr"][warp["grid_index"][f]], warp["resolution"], warp["length"], warp["seg"][f], warp["madj"][f])
        ora.Integrate(d, sc["traj"][f])


# ---- frame-split merge: one checker for the loopback ranks on one GPU and for real RCCL ranks on several ------------------------------
def merge_blocks(G, per, devices):
    """G contiguous blocks of one revolution through the 6 m room with the radius drifting (the configs[3] shape): neighbouring blocks share units at
    their borders, most units belong to one block.  Block r is rendered on devices[r]."""
    import torch
    from elasticreconstruction_amd import synth
    out = []
    for r in range(G):
        sc = synth.make_scenario(per, interval=50, warp=True, frame_offset=r * per, total_frames=G * per, revolutions=1.0, radius_drift=1.5,
                                 room=(-1.5, 4.5), device="cuda:%d" % devices[r])
        out.append((sc, synth.warp_arrays(sc)))
        torch.cuda.synchronize(devices[r])                     # rendered on torch's stream: finished before the library's streams read them
    return out


def band_record_words(sdf, w):
    """Size of a unit's band record (csrc/er_tsdf.hip, "band records"): header + weights of the observed voxels (16 bits each when every weight is a frame
    count below 65536) + sdf of the observed voxels whose sdf is not exactly 1, both padded to an even number of 32-bit words."""
    import numpy as np
    on = w != 0
    obs = int(np.count_nonzero(on))
    band = int(np.count_nonzero(on & (sdf.view(np.uint32) != 0x3f800000)))
    wo = w[on]
    wide = bool(obs and ((wo < 1).any() or (wo > 65535).any() or (wo != np.floor(wo)).any()))
    return 16644 + (((obs + 1) & ~1) if wide else 2 * ((obs + 3) // 4)) + ((band + 1) & ~1)


def band_record_bytes(words):
    return 4 * int(words)


def check_frame_split_merge(make_comms, devices, root, impl="owner", per=100, max_units=2048, repeat=1):
    """len(devices) ranks integrate their blocks into volumes of their own, er_tsdf_allreduce merges them (root >= 0, -1 = on every rank, -2 = left
    distributed), and the result is compared with ONE volume that integrated the same blocks in order:
      * key sets: the union, nothing lost, (distributed) every unit on exactly one rank -- a toucher that observed most of it;
      * units one rank touched: BIT-IDENTICAL to the single volume (no frame of another rank ever reached them);
      * units two or more ranks touched: weights exact, sdf within 1e-5 of the single volume (TSDFVolume.cpp:93-94 summed in another order) and, for the
        owner merge, BIT-IDENTICAL to the float32 sum of the ranks' volumes in rank order (numpy restatement): the order is a function of the key sets;
      * er_comm_merge_stats / _owner equal to the key-set and band arithmetic; what crosses the transport is records, not planes;
      * repeat > 1: the merge is run again from scratch and must give the same bits (bit-reproducibility at N > 1).
    Returns a summary dict."""
    import os
    import numpy as np
    import torch
    from elasticreconstruction_amd import parallel
    from elasticreconstruction_amd.tsdf import TSDFVolume
    G = len(devices)
    blocks = merge_blocks(G, per, devices)
    full = TSDFVolume(max_units=max_units, device=devices[0])
    for r, (sc, w) in enumerate(blocks):
        depth = sc["depth"] if devices[r] == devices[0] else sc["depth"].to("cuda:%d" % devices[0])
        torch.cuda.synchronize(devices[0])
        full.IntegrateFrames(None, sc["traj"], w, device_ptr=depth.data_ptr())
        full.synchronize()
    full_keys = [int(k) for k in full.unit_keys()]
    old = os.environ.get("ER_MERGE_IMPL")
    os.environ["ER_MERGE_IMPL"] = impl
    summary, previous = {}, None
    try:
        for rep in range(repeat):
            vols = [TSDFVolume(max_units=max_units, device=devices[r]) for r in range(G)]
            for v, (sc, w) in zip(vols, blocks):
                v.IntegrateFrames(None, sc["traj"], w, device_ptr=sc["depth"].data_ptr())
                v.synchronize()
            before = [{int(k): v.read_unit(int(k)) for k in v.unit_keys()} for v in vols]
            touch = {}
            for r, b in enumerate(before):
                for k in b:
                    touch.setdefault(k, []).append(r)
            multi = sorted(k for k, t in touch.items() if len(t) >= 2)
            single = sorted(k for k, t in touch.items() if len(t) == 1)
            assert sorted(touch) == full_keys
            assert len(multi) > 20 and len(single) > 100
            count = {(k, r): band_record_words(*before[r][k]) for k, t in touch.items() for r in t}      # (the protocol's "count" of a unit: its record size)
            for r in range(G):                                    # er_tsdf_band_sizes against the read-back volumes
                ks = sorted(before[r])
                assert [int(c) for c in vols[r].band_sizes(ks)] == [count[(k, r)] for k in ks]
            comms = make_comms()
            nu = comms.allreduce(vols, root=root)
            assert nu == len(touch)
            st = [comms.merge_stats(r) for r in range(G)]
            have = [set(int(k) for k in v.unit_keys()) for v in vols]
            # expected bits of the multi-toucher units: float32 sums in rank order
            def rank_sum(k):
                SW = np.zeros(64 ** 3, np.float32)
                W = np.zeros(64 ** 3, np.float32)
                for r in touch[k]:
                    s1, w1 = before[r][k]
                    SW = SW + s1 * w1                              # (an unobserved voxel adds +0: the same bits as skipping it)
                    W = W + w1
                with np.errstate(divide="ignore", invalid="ignore"):
                    return np.where(W > 0, SW / W, np.float32(0)).astype(np.float32), W
            if impl == "owner":
                for r in range(G):
                    assert st[r]["impl"] == "owner" and st[r]["bytes_reduced"] == 0
                    assert (st[r]["union_units"], st[r]["multi_toucher_units"], st[r]["single_toucher_units"]) == (len(touch), len(multi), len(single))
                    assert st[r]["ring_equivalent_bytes"] == len(multi) * 2 * 64 ** 3 * 4
                # the owners, read off the distributed result or recomputed: a toucher with the largest band
                best = {k: max(count[(k, r)] for r in touch[k]) for k in multi}
                if root == parallel.MERGE_DISTRIBUTED:
                    owner = {}
                    for k in touch:
                        holders = [r for r in range(G) if k in have[r]]
                        assert len(holders) == 1, "unit %d lives on ranks %s after a distributed merge" % (k, holders)
                        owner[k] = holders[0]
                        assert owner[k] in touch[k] and (len(touch[k]) == 1 or count[(k, owner[k])] == best[k])
                    to_owner = sum(band_record_bytes(count[(k, r)]) for k in multi for r in touch[k] if r != owner[k])
                    assert sum(s["to_owners_bytes_sent"] for s in st) == to_owner == sum(s["to_owners_bytes_received"] for s in st)
                    assert sum(s["to_root_bytes_sent"] for s in st) == 0
                    for r in range(G):
                        assert st[r]["units_owned"] == len(have[r]) and st[r]["units_summed_here"] == len([k for k in multi if owner[k] == r])
                        assert st[r]["units_handed_over"] == len([k for k in multi if r in touch[k] and owner[k] != r])
                    summary["to_owners_MB"] = to_owner / 1e6
                else:
                    to_owner = sum(s["to_owners_bytes_sent"] for s in st)
                    lo = sum(band_record_bytes(count[(k, r)]) for k in multi for r in touch[k]) - sum(band_record_bytes(best[k]) for k in multi)
                    assert to_owner == lo == sum(s["to_owners_bytes_received"] for s in st)
                    assert sum(s["to_root_bytes_sent"] for s in st) > 0
                    assert sum(s["to_root_bytes_received"] for s in st) == sum(s["to_root_bytes_sent"] for s in st)      # (sent counts every receiver)
                summary["ring_equivalent_MB"] = st[0]["ring_equivalent_bytes"] / 1e6
                summary["moved_MB"] = sum(s["bytes_sent"] for s in st) / 1e6
            else:
                unit_bytes = 2 * 64 ** 3 * 4
                for r in range(G):
                    assert st[r]["impl"] == "ring" and st[r]["bytes_reduced"] == len(multi) * unit_bytes
                    mine = [k for k in single if touch[k] == [r]]
                    travels = (lambda k: True) if root < 0 else (lambda k: touch[k] != [root])
                    assert st[r]["units_sent"] == len([k for k in mine if travels(k)])
                    want_recv = len([k for k in single if touch[k] != [r]]) if (root < 0 or r == root) else 0
                    assert st[r]["units_received"] == want_recv and st[r]["bytes_received"] == want_recv * unit_bytes
            receivers = list(range(G)) if root == parallel.MERGE_ALL else ([root] if root >= 0 else [])
            for r in receivers:
                assert sorted(have[r]) == full_keys, "rank %d misses units after the merge" % r
            assert set().union(*have) == set(full_keys)
            worst, bits = 0.0, {}
            for r in range(G):
                for k in sorted(have[r]):
                    if root >= 0 and r != root and impl == "owner" and k not in before[r]:
                        continue
                    if root >= 0 and r != root and impl != "owner":
                        continue                                  # (ring: the other ranks keep their partial volumes)
                    sm, wm = vols[r].read_unit(k)
                    sf, wf = full.read_unit(k)
                    if len(touch[k]) == 1:
                        assert np.array_equal(wf, wm) and np.array_equal(sf.view(np.uint32), sm.view(np.uint32)), "single-toucher unit %d changed (rank %d)" % (k, r)
                    elif root >= 0 and r != root:
                        continue                                  # (a non-root owner's copy is checked through the root's)
                    else:
                        assert np.array_equal(wf, wm), "merged weights differ in unit %d" % k
                        worst = max(worst, float(np.abs(sf - sm).max()))
                        if impl == "owner":
                            es, ew = rank_sum(k)
                            assert np.array_equal(ew, wm) and np.array_equal(es.view(np.uint32), sm.view(np.uint32)), "unit %d is not the rank-ordered float32 sum (rank %d)" % (k, r)
                    if len(touch[k]) > 1:
                        bits.setdefault(k, sm.view(np.uint32).copy())
                        assert np.array_equal(bits[k], sm.view(np.uint32)), "unit %d differs between ranks" % k
            assert worst <= 1e-5, "merged tsdf differs by %.3g" % worst
            if previous is not None:
                assert sorted(previous) == sorted(bits) and all(np.array_equal(previous[k], bits[k]) for k in bits), "the merge is not bit-reproducible"
            previous = bits
            summary.update({"union": len(touch), "multi": len(multi), "single": len(single), "max_abs_dsdf": worst, "root": root, "impl": impl, "ranks": G})
            comms.close()
            for v in vols:
                v.close()
    finally:
        if old is None:
            os.environ.pop("ER_MERGE_IMPL", None)
        else:
            os.environ["ER_MERGE_IMPL"] = old
        full.close()
    return summary
