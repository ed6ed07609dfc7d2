#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X: depth frames/s integrated into a 512^3 TSDF
(8x8x8 volume units of 64^3, 640x480 frames) with the control-grid warp on (config 2).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one fragment of the trajectory = --interval (50) consecutive depth frames handed to the hot
path in ONE call: Reproject (control-grid warp) + ScaleDepth + unit touch + IntegrateVolumeUnit, i.e.
what CIntegrateApp::Execute does for each of those frames (IntegrateApp.cpp:217-225).  Frames are
synthetic (elasticreconstruction_amd/synth.py), rendered straight into HBM before the clock starts.
Warm-up steps run on a scratch volume; the K timed steps fill a fresh one.

N > 1 (weak scaling): every rank integrates its own contiguous K*interval-frame block of one long
trajectory into a private volume (no data-path collective), then -- inside the timed region -- the
per-GPU volumes are merged by ONE RCCL reduce(sum) to rank 0 of the sdf*weight / weight planes of the union
of touched units (`parallel.merge_volumes(mode="all_reduce")` leaves the result on every rank).

Output: one JSON line (rank 0).  roofline.achieved uses the ALGORITHMIC bytes of SURVEY.md 8d,
B_A = 16 * N_upd + 1 843 200 (+ 1 228 800 with the warp) per frame with sum(N_upd) = sum(weight_),
divided by the average duration of the dominant kernel (k_integrate) measured with HIP events on
the stream it runs on.  cpu_baseline times the REFERENCE's own code (oracle/_ref, built from
/root/reference/Integrate/*.cpp unmodified, 8 OpenMP threads as hard-coded there) on a bounded
sample of the same frames; when that build is absent it falls back to the oracle port and says so.
"""
import argparse
import ctypes
import json
import os
import sys
import tempfile
import time

# OPT-IN, before the HIP runtime initialises (torch.cuda below): one hardware queue per stream of the TSDF pipeline (what
# er_request_hw_queues / elasticreconstruction_amd.request_hw_queues do; include/er_hip.h).  Never overrides the user's value.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("ER_ORACLE_QUIET", "1")          # the reference's own code logs through printf / cout: stdout carries ONE JSON line

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

C_void = ctypes.c_void_p
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # ... and the measured copy ceiling of this part ("report against both")
VALU_CLOCK_GHZ = 2.4           # peak engine clock ASSUMED for valu_issue.peak (the chip holds ~1.9-2.1 GHz under this load)
FRAME_BYTES_RAW = 640 * 480 * 2
CONFIG2_FRAMES = 3000          # BASELINE.json configs[1]
FRAME_BYTES_FIXED = 640 * 480 * (2 + 4)        # raw read + scaled write/read once


from bench_dry import DryComm, DryVolume
from bench_extras import (allpairs_section, boundary_section, cpu_baseline, fopt_section, icp_section, other_configs, parity_check,
                          sampled_parity)


JOB_FRAMES = {2: None, 4: 10000, 5: 5000}


def kernel_source_sha16():
    """sha256 over the sources of path A's kernels and their build flags (csrc/er_tsdf.hip + er_tsdf_math.h + Makefile), first 16 hex
    digits: stamped next to the static counter figures of profiles/pmc_latest.json so that a kernel change after the profiled run shows
    (VERDICT round 3, weak 8)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("er_tsdf.hip", "er_tsdf_math.h", "Makefile"):
        with open(os.path.join(ROOT, "elasticreconstruction_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def plan_steps(config, steps, frames_per_step, interval, world):
    """(K, S) = timed steps per rank and frames per step.  configs[1] (weak scaling): K = --steps, S = as many whole fragments as it
    takes for K steps to cover its 3000 frames.  configs[3] / configs[4] (strong scaling): every rank takes ceil(job / world) frames
    rounded up to whole fragments, a step = g fragments with g the first of 4, 5, 3, 2, 1 that divides the rank's fragment count, so
    that K * S is EXACTLY the rank's share (the dry run of round 4 found the old "K = job // (world * 200)" dropping 400 of
    configs[3]'s 10 000 frames at 8 ranks)."""
    K, S, I = steps, frames_per_step, interval
    job_frames = JOB_FRAMES[config]
    if job_frames:
        frags = -(-job_frames // (world * I))
        if S <= 0:
            S = I * next(g for g in (4, 5, 3, 2, 1) if frags % g == 0)
        if S % I or (frags * I) % S:
            raise SystemExit("--frames-per-step %d does not divide this rank's %d frames into whole fragments" % (S, frags * I))
        return frags * I // S, S
    if S <= 0:
        S = max(I, (CONFIG2_FRAMES // max(K, 1)) // I * I)     # whole fragments per step; K steps cover configs[1] when K divides 60
    if S % I:
        raise SystemExit("--frames-per-step must be a multiple of --interval")
    return K, S


COMPACT_LIMIT = 4096           # bytes: the driver keeps an 8 KB tail of stdout; BENCH_r05's 22 KB line was cut and parsed as null


def _sig(x, digits=8):
    """Numbers to `digits` significant figures (floats only), recursively: the compact line is for a parser, not for archaeology."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float("%.*g" % (digits, x)) if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return _sig(float(x), digits)


def _get(d, *path):
    for p in path:
        if not isinstance(d, dict) or d.get(p) is None:
            return None
        d = d[p]
    return d


def compact_line(out):
    """The ONE stdout line: numbers and short identifiers only.  Everything else (definitions, per-phase tables, the children's
    full objects) lives in bench_full.json.  roofline.frac is SURVEY.md 8d's contract figure and recomputes from this line:
    algorithmic_bytes_per_pass / (ms_per_step x steps x 1e-3) / 8e12; kernel_frac / achieved / avg_launch_ms price k_integrate
    with its own 16 B per voxel update over its HIP-event launch time."""
    rf, cfg = out.get("roofline") or {}, out.get("config") or {}
    c = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data")}
    if out.get("dry_run"):
        c["dry_run"] = True
    c["config"] = {"workload": (cfg.get("workload") or "").split(":")[0], "baseline_config": cfg.get("baseline_config"),
                   "frames_per_step": cfg.get("frames_per_step"), "frames_per_gpu": cfg.get("frames_per_gpu"),
                   "volume_units_touched": cfg.get("volume_units_touched"), "warp": cfg.get("warp"),
                   "inputs": "host" if (cfg.get("inputs") or "").startswith("HOST") else "hbm_resident",
                   "parallelism": cfg.get("parallelism")}
    for k in ("rccl_ranks", "merge_union_units"):
        if k in cfg:
            c["config"][k] = cfg[k]
    if "merge_impl" in cfg:
        c["config"]["merge_impl"] = cfg["merge_impl"].split(" ")[0]
    ms = _get(cfg, "merge_stats")
    if isinstance(ms, dict):
        c["config"]["merge"] = {k: ms[k] for k in ("impl", "multi_toucher_units", "single_toucher_units", "bytes_sent", "bytes_received", "bytes_reduced",
                                                   "ring_equivalent_bytes") if k in ms}
        c["config"]["merge_result"] = (cfg.get("merge_result") or "").split(" ")[0]
    kint = _get(rf, "kernels", "k_integrate") or {}
    c["roofline"] = {"bound": rf.get("bound"), "kernel": rf.get("kernel"), "frac": rf.get("frac"), "achieved": rf.get("achieved"),
                     "peak": rf.get("peak"), "unit": rf.get("unit"), "algorithmic_bytes_per_pass": rf.get("algorithmic_bytes_per_pass"),
                     "kernel_frac": rf.get("kernel_frac"), "kernel_frac_rocprof": rf.get("kernel_frac_rocprof"),
                     "kernel_achieved": kint.get("achieved"), "kernel_bytes_per_launch": kint.get("algorithmic_bytes_per_launch"),
                     "avg_launch_ms": rf.get("avg_launch_ms"), "launches": rf.get("launches"),
                     "kernel_alone_frac": _get(rf, "kernel_alone", "frac"), "kernel_alone_ms": _get(rf, "kernel_alone", "avg_launch_ms"),
                     "traffic": rf.get("traffic"), "hbm_physical_frac": rf.get("hbm_physical_frac"),
                     "valu_issue_frac": _get(rf, "valu_issue", "frac"), "valu_job_frac": _get(rf, "valu_issue", "job_frac"),
                     "static_stale": _get(rf, "static_figures", "stale")}
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                             "sample": cb.get("sample_short") or (cb.get("sample") or "")[:60],
                             "uncapped": _get(cb, "uncapped", "value"), "uncapped_threads": _get(cb, "uncapped", "threads")}
    pc = out.get("parity_checked")
    if pc:
        c["parity_checked"] = {"bit_exact": pc.get("bit_exact"), "frames": pc.get("frames"), "units": pc.get("units_gpu"), "against": (pc.get("against") or "").split(" ")[0]}
    if out.get("streamed"):
        c["streamed"] = {"value": out["streamed"].get("value"), "h2d_GB_per_s": _get(out["streamed"], "h2d_copy_only", "GB_per_s")}
    i = out.get("icp")
    if i:
        ci = {"pairs_per_s": i.get("pairs_per_s"), "pairs": i.get("pairs", i.get("pairs_total")), "mean_icp_iterations": i.get("mean_icp_iterations"),
              "roofline_frac": _get(i, "roofline", "frac"), "bytes_per_pair": _get(i, "roofline", "algorithmic_bytes_per_pair"),
              "nn_queries_per_s": i.get("nn_queries_per_s"),
              "fused_pairs_per_s": _get(i, "fused_entry", "pairs_per_s"), "device_hand_off_pairs_per_s": _get(i, "device_hand_off", "pairs_per_s"),
              "incl_cloud_build_pairs_per_s": i.get("pairs_per_s_incl_cloud_build_batched"),
              "hard_pairs_per_s": _get(i, "hard_set", "pairs_per_s"), "hard_ref_ok": _get(i, "hard_set", "parity_checked_reference", "ok"),
              "realistic_pairs_per_s": _get(i, "realistic", "pairs_per_s"), "realistic_ratio": _get(i, "realistic", "ratio_to_the_uniform_list"),
              "realistic_ref_ok": _get(i, "realistic", "parity_checked_reference", "ok"),
              "parity_ok": _get(i, "parity_checked", "ok"), "parity_ref_ok": _get(i, "parity_checked_reference", "ok"),
              "max_abs_T_diff": _get(i, "parity_checked", "max_abs_T_diff"),
              "ransac_hypotheses_per_s": _get(i, "ransac_fitness", "hypotheses_per_s")}
        if i.get("cpu_baseline"):
            b = i["cpu_baseline"]
            ci["cpu_baseline"] = {"value": b.get("value"), "unit": b.get("unit"), "cores": b.get("cores"), "kind": b.get("kind"), "pairs": b.get("pairs"),
                                  "uncapped": _get(b, "uncapped", "value")}
        for k in ("pairs_total", "accepted_this_rank", "rejected_by_pre_check", "sharding"):
            if k in i:
                ci[k] = i[k]
        c["icp"] = {k: v for k, v in ci.items() if v is not None}
    fo = out.get("fragment_optimizer")
    if fo:
        c["fragment_optimizer"] = {k: fo.get(k) for k in ("slac_assembly_ms", "slac_correspondences_per_s", "cpu_port_slac_correspondences_per_s")}
    oc = out.get("other_configs")
    if oc:
        c["other_configs"] = {}
        for name, r in oc.items():
            if not isinstance(r, dict):
                continue
            if "error" in r:
                c["other_configs"][name] = {"error": str(r["error"])[:80]}
                continue
            e = {"value": r.get("value"), "ms_per_step": r.get("ms_per_step"), "steps": r.get("steps"), "frames": r.get("frames"),
                 "units": r.get("volume_units_touched"), "roofline_frac": r.get("roofline_frac"),
                 "bit_exact": _get(r, "parity_checked", "bit_exact"), "cpu_value": _get(r, "cpu_baseline", "value")}
            if r.get("scaling") == "strong" and r.get("rccl_ranks"):
                e.update({"scaling": "strong", "rccl_ranks": r.get("rccl_ranks"), "merge_bytes_sent": _get(r, "merge_stats", "bytes_sent")})
            if r.get("icp"):
                e["icp_pairs_per_s"] = r["icp"].get("pairs_per_s")
                e["icp_parity_ok"] = _get(r["icp"], "parity_checked", "ok")
            c["other_configs"][name] = e
    bd = out.get("boundary")
    if bd:
        c["boundary"] = bd.get("compact", bd)
    c["full"] = "bench_full.json"
    line = json.dumps(_sig(c), separators=(",", ":"))
    if len(line) >= COMPACT_LIMIT:                 # never lose the headline over an appendix: drop the optional objects, largest first
        for k in ("fragment_optimizer", "boundary", "streamed", "other_configs", "icp"):
            c.pop(k, None)
            line = json.dumps(_sig(c), separators=(",", ":"))
            if len(line) < COMPACT_LIMIT:
                break
    return line


class QuietStdout:
    """File descriptor 1 points at stderr for the whole run and comes back for the ONE line: RCCL prints a version banner through C stdio (seen
    in the first round-6 GPU run, flushed at process exit, i.e. AFTER the JSON line), the reference's code logs through printf / cout, and with
    N > 1 every rank's stdout lands in the same pipe.  Ranks other than 0 never get it back."""

    def __init__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def restore(self):
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)                 # whatever C stdio still holds goes where fd 1 points NOW (stderr)
        except OSError:
            pass
        os.dup2(self._saved, 1)


_quiet = None


def emit(out, full_path):
    """Full object -> file (and stderr), compact line -> stdout, LAST."""
    try:
        with open(full_path, "w") as fh:
            json.dump(out, fh)
            fh.write("\n")
        god = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(god) and os.path.dirname(os.path.abspath(full_path)) == ROOT:
            with open(os.path.join(god, "bench_full.json"), "w") as fh:
                json.dump(out, fh)
    except OSError as ex:
        sys.stderr.write("bench.py: could not write %s: %s\n" % (full_path, ex))
    sys.stderr.write("bench.py full result object:\n" + json.dumps(out, indent=1) + "\n")
    sys.stderr.flush()
    if _quiet is not None:
        _quiet.restore()
    print(compact_line(out), flush=True)
    if _quiet is not None:
        os.dup2(2, 1)                                      # (anything a library prints at exit stays off the line's pipe)



def strong_scaling_leg(args, dist, dev, local, rank, world, comm, dry, stream):
    """BASELINE.json configs[3] INSIDE an N > 1 run: the 10 000-frame job of the drifting path cut into `world` contiguous blocks,
    every rank its block into a private volume, the merge (by unit owner; result distributed) inside the timed region -- the strong-scaling line of
    the same communicator the headline just used, so that the driver's SCALE file carries the curve that matters.  Returns the object rank 0 attaches
    as other_configs['configs[3]'] (None on the other ranks)."""
    import numpy as np
    import torch
    from elasticreconstruction_amd import parallel, synth
    from elasticreconstruction_amd.tsdf import TSDFVolume
    I = args.interval
    K, S = plan_steps(4, args.steps, 0, I, world)
    n_frames = K * S
    sc = synth.make_scenario(n_frames, interval=I, warp=True, frame_offset=rank * n_frames, total_frames=world * n_frames,
                             revolutions=max(1.0, world * n_frames / float(CONFIG2_FRAMES)), radius_drift=1.5, room=(-1.5, 4.5), device=dev)
    depth, px = sc["depth"], sc["depth"].shape[1]
    warp_all = synth.warp_arrays(sc)
    vol = DryVolume(rank, 4096) if dry else TSDFVolume(max_units=4096, device=local)
    vol.set_stream(stream.cuda_stream if stream is not None else None)
    sync = (lambda: None) if dry else torch.cuda.synchronize
    root = 0 if dry else args.merge_root

    def one_pass():
        vol.reset()
        sync()
        dist.barrier()
        sync()
        t0 = time.perf_counter()
        for s in range(K):
            lo, hi = s * S, (s + 1) * S
            gi = warp_all["grid_index"][lo:hi]
            g0, g1 = int(gi.min()), int(gi.max()) + 1
            w = dict(ctr=warp_all["ctr"][g0:g1], resolution=warp_all["resolution"], length=warp_all["length"], grid_index=gi - g0,
                     seg=warp_all["seg"][lo:hi], madj=warp_all["madj"][lo:hi])
            vol.IntegrateFrames(None, sc["traj"][lo:hi], w, device_ptr=depth.data_ptr() + lo * px * 2)
        nu = comm.allreduce(vol, root=root) if comm is not None else parallel.merge_volumes(vol, dist, dev)
        sync()
        dist.barrier()
        sync()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), nu
    one_pass()                                                           # warm-up: buffers of the merge, the unit pool
    passes, nu = [], 0
    for _ in range(3):
        dt, nu = one_pass()
        passes.append(dt)
    dt = float(np.median(passes))
    distributed = comm is not None and not dry and root == -2
    t = torch.tensor([vol.sum_weight() if (distributed or rank == max(root, 0)) else 0.0], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    sum_w = float(t.item())
    stats = comm.merge_stats() if comm is not None and hasattr(comm, "merge_stats") else None
    vol.close()
    if rank != 0:
        return None
    total = world * n_frames
    bytes_pass = 16.0 * sum_w + (FRAME_BYTES_FIXED + 2 * FRAME_BYTES_RAW) * total
    return {"value": total / dt, "unit": "frames/s", "scaling": "strong", "ms_per_step": 1e3 * dt / K, "steps": K, "frames": total, "frames_per_gpu": n_frames,
            "volume_units_touched": int(nu), "roofline_frac": bytes_pass / dt / 1e9 / HBM_PEAK_GBS / world, "roofline_frac_definition": "per GPU: job bytes / time / (n_gpus x 8 TB/s)",
            "pass_ms": [round(1e3 * p, 3) for p in passes], "merge_stats": stats, "rccl_ranks": world,
            "workload": "configs[3]: 10 000 frames of the drifting path in %d contiguous blocks, merge by unit owner inside the timed region" % world}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=2, choices=[2, 4, 5],
                    help="BASELINE.json config (1-based): 2 = the headline (3000 frames, 512^3, warp; the default and what the driver "
                         "runs); 4 = 10 000 frames on a drifting path through a 6 m room into the hashed unit grid (> 512 units), frame "
                         "blocks per rank + the RCCL merge (run on one GPU it exercises the same code at G = 1); 5 = the 100-fragment "
                         "scene: all 4950 pairs through BuildCorrespondence's flow, then 5000 frames integrated")
    ap.add_argument("--interval", type=int, default=50, help="frames per fragment / control grid (--interval of Integrate)")
    ap.add_argument("--frames-per-step", type=int, default=0,
                    help="frames handed to the hot path per step (one er_tsdf_integrate_frames call); 0 = as many whole fragments "
                         "as it takes for --steps steps to cover all %d frames of configs[1] (150 at the default 20 steps)" % CONFIG2_FRAMES)
    ap.add_argument("--min-seconds", type=float, default=2.5,
                    help="repeat the K-step pass on an emptied volume until the passes add up to this much timed work (>= 2 s, so that a "
                         "utilisation sampler sees the GPU leg of the run)")
    ap.add_argument("--max-passes", type=int, default=400)
    ap.add_argument("--no-warp", action="store_true", help="rigid --ref_traj style run (no control grid)")
    ap.add_argument("--cpu-sample", type=int, default=200, help="frames timed on the CPU reference (0 = skip)")
    ap.add_argument("--host-input", action="store_true",
                    help="hand the depth frames over as (pageable) HOST memory in the headline loop; never the headline configuration")
    ap.add_argument("--event-stride", type=int, default=4,
                    help="bracket every n-th k_integrate launch of the timed passes with HIP events (1 = every launch: costs ~2 %% of the rate; 0 = none)")
    ap.add_argument("--no-alone", action="store_true", help="skip the extra untimed pass that runs k_integrate alone (profiler runs: keeps "
                                                            "the kernel-trace average comparable with roofline.avg_launch_ms)")
    ap.add_argument("--no-streamed", action="store_true", help="skip the extra pass that streams the frames from page-locked host memory")
    ap.add_argument("--force-merge", action="store_true",
                    help="run the frame-split merge (key all-gather + all-reduce) even with one rank: exercises the RCCL path on a 1-GPU box")
    ap.add_argument("--merge-impl", choices=["torch", "abi"], default="abi",
                    help="frame-split merge through liber_hip.so's own RCCL calls (er_tsdf_allreduce: the product path, what "
                         "bin/Integrate --gpus uses; default) or through torch.distributed (parallel.merge_volumes, the cross-check)")
    ap.add_argument("--merge-root", type=int, default=-2,
                    help="where the frame-split merge leaves the result (--merge-impl abi): -2 = distributed by unit owner (default, what bin/Integrate "
                         "--gpus N does), r >= 0 = gathered on rank r, -1 = on every rank")
    ap.add_argument("--icp-pairs", type=int, default=50,
                    help="also time N fragment pairs per GPU through Registration + FindCorrespondence (configs[2] shape) and add an "
                         "'icp' object with BASELINE.json's second figure, pairs/s (0 = skip)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse this script's control flow WITHOUT a GPU: gloo instead of RCCL, a host-array stand-in for the volume "
                         "(DryVolume), no ICP / CPU-baseline legs.  Nothing is measured -- the JSON line says dry_run: true and its numbers "
                         "are meaningless; tests/test_distributed_cpu.py runs it with 4 ranks so that the first 8-GPU run of the driver "
                         "cannot die on argument handling, rank gating, the collective sequence or the JSON shape")
    ap.add_argument("--other-configs", type=int, default=1,
                    help="(default run only: --config 2, one GPU, frames resident) also run 'bench.py --config 4' and '--config 5' as child "
                         "processes after the headline measurement and attach a summary of their JSON lines as 'other_configs' (0 = skip)")
    ap.add_argument("--boundary", type=int, default=1,
                    help="(default run only) also time the drop-in PROGRAMS end to end -- bin/Integrate on configs[1] from files, bin/BuildCorrespondence on "
                         "the 50-pair list -- beside the reference's own programs on a bounded sample (0 = skip)")
    ap.add_argument("--scaling-child", type=int, default=1,
                    help="(--gpus N > 1, --config 2) after the weak-scaling headline also run configs[3]'s 10 000-frame job cut into N blocks over the same "
                         "communicator and attach it as other_configs['configs[3]'] (0 = skip)")
    ap.add_argument("--full-json", default=os.path.join(ROOT, "bench_full.json"),
                    help="where rank 0 writes the FULL result object (per-phase tables, definitions, A/B leftovers, the children's objects); "
                         "stdout carries only the compact line (< 4 KB) made from it by compact_line()")
    args = ap.parse_args()
    global _quiet
    _quiet = QuietStdout()

    import numpy as np
    import torch
    import torch.distributed as dist
    from elasticreconstruction_amd import synth
    from elasticreconstruction_amd.tsdf import TSDFVolume

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    dry = args.dry_run
    if dry:
        dev = torch.device("cpu")
        torch.set_num_threads(2)
        cuda_sync = lambda: None
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        cuda_sync = torch.cuda.synchronize
    use_dist = world > 1 or args.force_merge or args.config == 4
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    K, W, I = args.steps, args.warmup, args.interval
    S = args.frames_per_step
    job_frames = JOB_FRAMES[args.config]                             # configs 4 / 5 fix the JOB's frame count (strong scaling)
    K, S = plan_steps(args.config, K, S, I, world)
    n_frames = K * S
    warp_on = not args.no_warp
    # one long trajectory split into contiguous per-rank blocks (config 4's frame-batch shard)
    big_room = args.config == 4
    if dry:                                                # the stand-in volume never looks at a pixel: no rendering on the CPU
        synth.render_depth = lambda w, lo=None, hi=None, device="cpu": torch.zeros((len(w), 640 * 480), dtype=torch.int16, device=device)
    sc = synth.make_scenario(n_frames, interval=I, warp=warp_on, frame_offset=rank * n_frames,
                             total_frames=world * n_frames, revolutions=max(1.0, world * n_frames / float(CONFIG2_FRAMES)),
                             radius_drift=1.5 if big_room else 0.0, room=(-1.5, 4.5) if big_room else (synth.ROOM_LO, synth.ROOM_HI),
                             device=dev)
    depth = sc["depth"]                                    # uint16 [n_frames, 307200] in HBM
    warp_all = synth.warp_arrays(sc) if warp_on else None

    def warp_slice(lo, hi):
        if not warp_on:
            return None
        gi = warp_all["grid_index"][lo:hi]
        g0, g1 = int(gi.min()), int(gi.max()) + 1                       # only the grids this step needs travel
        return dict(ctr=warp_all["ctr"][g0:g1], resolution=warp_all["resolution"], length=warp_all["length"],
                    grid_index=gi - g0, seg=warp_all["seg"][lo:hi], madj=warp_all["madj"][lo:hi])

    # A dedicated (non-null) torch stream carries torch ops, RCCL ordering AND every kernel of the handle,
    # so HIP-event timing and the all-reduce see one in-order queue.
    stream = None
    if not dry:
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.synchronize()
        torch.cuda.set_stream(stream)
    px = depth.shape[1]

    depth_host = synth.to_numpy_u16(depth) if args.host_input else None

    def run_steps(vol, k, host=None):
        for s in range(k):
            lo, hi = s * S, (s + 1) * S
            if host is not None:
                vol.IntegrateFrames(host[lo:hi], sc["traj"][lo:hi], warp_slice(lo, hi))
            else:
                vol.IntegrateFrames(None, sc["traj"][lo:hi], warp_slice(lo, hi), device_ptr=depth.data_ptr() + lo * px * 2)

    comm = None
    merge_note = None
    if use_dist and args.merge_impl == "abi":
        from elasticreconstruction_amd import parallel
        try:
            comm = DryComm(dist, dev) if dry else parallel.AbiComm(dist, local)
            failed = 0
        except Exception as ex:                                          # e.g. librccl.so.1 not loadable from the library
            comm, failed, merge_note = None, 1, "er_comm_create failed on rank %d: %s" % (rank, ex)
        flag = torch.tensor([failed], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)                      # the choice of implementation must be the same on every rank
        if int(flag.item()):
            if comm is not None:
                comm.close()
            comm = None
            merge_note = merge_note or "er_comm_create failed on another rank"
            args.merge_impl = "torch"

    # the product's merge (er_tsdf_allreduce, round 6: reduce-scatter by unit owner) leaves the merged volume DISTRIBUTED -- every unit complete on
    # exactly one rank, which is all SaveWorld needs (bin/Integrate --gpus N assembles world.pcd from the ranks' extractions); --merge-root 0 gathers
    # it on rank 0 inside the timed region instead.  The torch cross-check and the dry run know the rooted form only.
    distributed = comm is not None and not dry and args.merge_root == -2

    def merge(vol):
        """Frame-split merge inside the timed region."""
        from elasticreconstruction_amd import parallel
        if comm is not None:
            return comm.allreduce(vol, root=args.merge_root if not dry else 0)
        return parallel.merge_volumes(vol, dist, dev)

    max_units = 4096 if big_room else (640 if world == 1 else 1024)
    vol = DryVolume(rank, max_units) if dry else TSDFVolume(max_units=max_units, device=local)
    vol.set_stream(stream.cuda_stream if stream is not None else None)
    # ---- warm-up (the volume is emptied afterwards) --------------------------------------------
    run_steps(vol, min(W, K), depth_host)
    if use_dist:
        merge(vol)                # also warms RCCL (communicator set-up, allocator) outside the timed region
    vol.synchronize()

    def timed_pass(host=None):
        """EXACTLY K steps into an emptied volume, bracketed by barrier + synchronize on both sides."""
        vol.reset()
        cuda_sync()
        if use_dist:
            dist.barrier()
        cuda_sync()
        t0 = time.perf_counter()
        run_steps(vol, K, host)
        nu = merge(vol) if use_dist else 0
        cuda_sync()
        if use_dist:
            dist.barrier()
        cuda_sync()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, nu

    vol.set_profiling(max(args.event_stride, 0))
    pass_s, n_union = [], 0
    while True:
        dt, n_union = timed_pass(depth_host)
        pass_s.append(dt)
        go = 1.0 if (sum(pass_s) < args.min_seconds and len(pass_s) < args.max_passes) else 0.0
        if use_dist:                                        # every rank must take the same decision
            t = torch.tensor([go], device=dev, dtype=torch.float64)
            dist.broadcast(t, src=0)
            go = float(t.item())
        if go == 0.0:
            break
    prof = vol.get_profile()
    vol.set_profiling(False)
    # k_integrate WITHOUT the overlap: one more (untimed) pass, one 50-frame launch at a time with a synchronise after each,
    # so the kernel runs alone on the chip.  In the timed passes it shares the SIMDs with the pre-passes of the next two
    # batches, which is faster for the job and slower for the kernel; both durations are reported.
    alone = None
    if rank == 0 and world == 1 and not args.host_input and not args.no_alone and not dry:
        vol.reset()
        vol.set_profiling(True)
        for lo in range(0, n_frames, I):
            vol.IntegrateFrames(None, sc["traj"][lo:lo + I], warp_slice(lo, lo + I), device_ptr=depth.data_ptr() + lo * px * 2)
            vol.synchronize()
        alone = vol.get_profile()
        vol.set_profiling(False)
    n_pass = len(pass_s)
    dt = float(np.median(pass_s))
    # unit weights add exactly: after a merge gathered on rank 0 it holds the job-wide number of voxel updates of ONE pass; after a distributed merge
    # every unit lives on exactly one rank and the ranks' sums add up to it (one small all-reduce, outside the timed region)
    sum_w = vol.sum_weight()
    n_units = vol.unit_count()
    if use_dist and distributed:
        t = torch.tensor([sum_w, float(n_units)], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        sum_w, n_units = float(t[0].item()), int(round(float(t[1].item())))
    sum_w /= world

    # ---- streamed: the same K steps with the frames in page-locked HOST memory (PCIe inside the timed region) ----
    streamed = None
    if rank == 0 and world == 1 and not args.no_streamed and not args.host_input and not dry:
        from elasticreconstruction_amd import _ffi
        arena = _ffi.PinnedArena()
        arena.reset(n_frames * px * 2 + 8192)
        pinned = arena.take((n_frames, px), np.uint16)
        pinned[...] = synth.to_numpy_u16(depth)
        timed_pass(pinned)
        ts = [timed_pass(pinned)[0] for _ in range(3)]
        # the PCIe ceiling of this box for the same bytes: one plain H2D copy of the whole pinned block
        tmp = torch.empty((n_frames, px), dtype=torch.int16, device=dev)
        tc = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _ffi.lib().er_host_copy_h2d(C_void(tmp.data_ptr()), C_void(pinned.ctypes.data), pinned.nbytes)
            tc.append(time.perf_counter() - t0)
        del tmp
        streamed = {"value": n_frames / float(np.median(ts)), "unit": "frames/s", "passes": 3,
                    "h2d_copy_only": {"frames_per_s": n_frames / float(np.median(tc)), "GB_per_s": pinned.nbytes / float(np.median(tc)) / 1e9,
                                      "what": "hipMemcpy of the same page-locked frames with no kernel running: this box's PCIe ceiling"},
                    "what": "same K steps, depth frames handed over as page-locked HOST memory (er_host_alloc): H2D copies on their "
                            "own stream overlap the pre-pass and the voxel pass; PCIe Gen5 x16 caps this near 100 k frames/s "
                            "(614 400 B per frame); never the headline value"}
        arena.close()

    icp = None
    if dry and (args.config == 5 or (args.icp_pairs > 0 and args.config == 2)):
        # the stand-in carries exactly the keys the gating below reads and rewrites
        icp = {"dry_run": True, "pairs_per_s": 1.0, "pairs": args.icp_pairs, "pairs_total": 4950, "nn_queries_per_s": 1.0, "_pass_s": 1.0 + 0.01 * rank}
    if dry and icp is not None:
        ap_s = icp_pass_s = icp.pop("_pass_s")
        if use_dist:
            t = torch.tensor([ap_s], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ap_s = icp_pass_s = float(t.item())
        if args.config == 5:
            icp["pairs_per_s"] = icp["pairs_total"] / ap_s
        elif use_dist:
            icp.update({"pairs_per_s": world * args.icp_pairs / icp_pass_s, "pairs": world * args.icp_pairs, "nn_queries_per_s": None,
                        "sharding": "%d GPUs x %d pairs, no collective; slowest rank's median pass" % (world, args.icp_pairs)})
    elif args.config == 5:
        icp = allpairs_section(100, local, rank, world, with_cpu=(rank == 0))
        ap_s = icp.pop("_pass_s")
        if use_dist:
            t = torch.tensor([ap_s], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ap_s = float(t.item())
        icp["pairs_per_s"] = icp["pairs_total"] / ap_s           # whole job: every rank's share done when the slowest is
    elif args.icp_pairs > 0 and args.config == 2:
        # secondary metric on every rank (pairs shard with no collective: each GPU runs the same pair list, weak scaling)
        icp = icp_section(args.icp_pairs, local, with_cpu=(rank == 0 and world == 1))
        icp_pass_s = icp.pop("_pass_s")
        if use_dist:
            t = torch.tensor([icp_pass_s], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            icp_pass_s = float(t.item())
            icp["pairs_per_s"] = world * args.icp_pairs / icp_pass_s
            icp["pairs"] = world * args.icp_pairs
            icp["nn_queries_per_s"] = None
            icp["sharding"] = "%d GPUs x %d pairs, no collective; slowest rank's median pass" % (world, args.icp_pairs)

    # N > 1: configs[3]'s strong-scaling job over the same communicator (every rank takes part; rank 0 gets the object)
    strong = None
    if world > 1 and args.config == 2 and args.scaling_child and not args.host_input:
        try:
            strong = strong_scaling_leg(args, dist, dev, local, rank, world, comm, dry, stream)
        except Exception as ex:                                          # (every rank raises or none does only by luck: keep the headline, say what happened)
            strong = {"error": repr(ex)[:300]} if rank == 0 else None
    if rank == 0:
        total_frames = world * n_frames
        out = {
            "metric": "depth frames/sec into 512^3 TSDF (640x480)",
            "value": total_frames / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1000.0 * dt / K,
            "higher_is_better": True,
            "scaling": "strong" if job_frames else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            **({"dry_run": True, "dry_run_note": "control-flow rehearsal on CPU (gloo, host-array volume): NOTHING here is a measurement"} if dry else {}),
            "config": {"workload": {2: "configs[1]", 4: "configs[3] (10 000 frames, 6 m room, drifting path, hashed unit grid)",
                                    5: "configs[4] (100-fragment scene: all-pairs ICP, then 5000 frames)"}[args.config] +
                                   ": %d synthetic 640x480 frames per GPU (box room + sphere, circular trajectory), "
                                   "TSDF units of 64^3 at 3/512 m, %s, %d frames per step"
                                   % (n_frames, "ControlGrid warp res 8 / %d grids" % (n_frames // I) if warp_on else "rigid", S),
                       "baseline_config": args.config, "warp": warp_on,
                       "frames_per_step": S, "frames_per_gpu": n_frames, "volume_units_touched": n_units,
                       "covers_all_of_configs1": bool(args.config == 2 and n_frames == CONFIG2_FRAMES),
                       "parallelism": "frame-block shard x%d + merge by unit owner" % world if world > 1 else "single GPU",
                       "inputs": "HOST memory, copied over PCIe inside the timed region (not the headline configuration)"
                       if args.host_input else "resident in HBM before the timed region"},
            "timing": {"passes": n_pass, "timed_region_s": float(sum(pass_s)), "pass_ms": {"min": 1e3 * min(pass_s), "median": 1e3 * dt, "max": 1e3 * max(pass_s)},
                       "what": "each pass = exactly K steps into an emptied volume (er_tsdf_reset outside the clock), barrier + synchronize "
                               "on both sides, max over ranks; value and ms_per_step come from the MEDIAN pass; passes repeat until "
                               "their sum reaches --min-seconds"},
        }
        if use_dist:
            out["config"]["merge_union_units"] = n_union
            out["config"]["merge_impl"] = args.merge_impl + (" (er_tsdf_allreduce: liber_hip.so's own RCCL calls)" if args.merge_impl == "abi" else " (parallel.merge_volumes over torch.distributed)")
            out["config"]["rccl_ranks"] = world
            out["config"]["merge_result"] = "distributed by unit owner" if distributed else "on rank %d" % max(args.merge_root, 0)
            if comm is not None and hasattr(comm, "merge_stats"):
                # rank 0's view of the LAST merge: the sum reduction carries only the units two or more ranks touched, the others travel raw, point to
                # point, or stay where they are (csrc/er_merge_protocol.h; with one rank nothing moves at all)
                out["config"]["merge_stats"] = comm.merge_stats()
            if merge_note:
                out["config"]["merge_impl_note"] = merge_note
        timed_launches = max(prof["launches"], 1)                 # every --event-stride-th launch of the timed passes carries HIP events
        ms_launch = prof["integrate_ms"] / timed_launches
        per_step = -(-S // 64)                                     # er_tsdf_integrate_frames: ceil(S / 64) launches of equal size per step
        launches = n_pass * K * per_step
        frames_per_launch = S / float(per_step)
        if prof["integrate_ms"] > 0:
            bytes_pass = 16.0 * sum_w + FRAME_BYTES_FIXED * n_frames + (2 * FRAME_BYTES_RAW * n_frames if warp_on else 0)
            per_launch = bytes_pass * n_pass / launches
            ach = per_launch / (ms_launch * 1e-3) / 1e9
            traffic, valu, phys, static, frac_rocprof, pj = None, None, None, None, None, {}
            pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
            if os.path.exists(pmc) and abs(frames_per_launch - 50.0) < 1e-9:
                try:
                    pj = json.load(open(pmc))
                    now = kernel_source_sha16()
                    static = {"file": "profiles/pmc_latest.json", "run": pj.get("run"), "kernel_source_sha16_at_that_run": pj.get("kernel_source_sha16"),
                              "kernel_source_sha16_now": now, "stale": pj.get("kernel_source_sha16") != now,
                              "what": "traffic, valu_issue.*wave_instructions* and frac_rocprof are STATIC figures of a committed rocprofv3 run, "
                                      "not of this run; sha16 = sha256 over csrc/er_tsdf.hip + er_tsdf_math.h + Makefile -- 'stale' means the kernels or their build flags changed since"}
                    traffic = pj.get("k_integrate_hbm_bytes_per_launch")                     # measured on a 50-frame launch
                    if traffic:
                        phys = traffic / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS
                    if pj.get("rocprof_kernel_trace_avg_us"):
                        frac_rocprof = per_launch / (float(pj["rocprof_kernel_trace_avg_us"]) * 1e-6) / 1e9 / HBM_PEAK_GBS
                    wi = pj.get("k_integrate_valu_wave_instructions_per_launch")
                    if wi:
                        # the limiter that actually binds: VALU issue.  peak = SIMDs x clock / 4 cycles per wave64 instruction
                        clk = float(pj.get("measured_clock_ghz") or VALU_CLOCK_GHZ)
                        peak = (256 if dry else torch.cuda.get_device_properties(local).multi_processor_count) * 4 * clk * 1e9 / 4.0
                        valu = {"wave_instructions_per_launch": wi, "achieved": wi / (ms_launch * 1e-3), "peak": peak,
                                "unit": "wave-instructions/s", "frac": wi / (ms_launch * 1e-3) / peak, "clock_ghz": clk,
                                "clock_source": "GRBM_GUI_ACTIVE / kernel duration of the same PMC run" if pj.get("measured_clock_ghz")
                                else "ASSUMED peak engine clock (not measured in this run)",
                                "source": "static: SQ_INSTS_VALU of %s (profiles/pmc_latest.json) over the LIVE launch time; "
                                          "peak = 1024 SIMDs x clock_ghz / 4 cycles per wave64 instruction" % pj.get("run", "a committed rocprofv3 --pmc run")}
                        per_batch = pj.get("valu_wave_instructions_per_batch")
                        if per_batch:
                            # ALL kernels of a batch (voxel pass + both pre-pass kernels) over the job's time per batch: the chip-wide figure
                            tot = float(sum(per_batch.values()))
                            batch_s = dt / (n_frames / frames_per_launch)
                            valu["job"] = {"wave_instructions_per_batch": per_batch, "total": tot, "batch_ms": 1e3 * batch_s,
                                           "achieved": tot / batch_s, "frac": tot / batch_s / peak,
                                           "what": "SQ_INSTS_VALU of every kernel of one 50-frame batch (static) over this run's time per batch: the share "
                                                   "of the chip's VALU issue peak the whole pipeline uses"}
                            valu["job_frac"] = tot / batch_s / peak
                except Exception:
                    traffic = None
            # ---- the contract figure: ALL algorithmic bytes of the pass over the timed wall time of the pass -------------
            job_ach = bytes_pass / dt / 1e9
            # ---- per kernel: each kernel priced with ONLY the bytes it is responsible for ------------------------------------------------
            n_launch_pass = launches / float(n_pass)                                  # launches of each kernel per pass
            kb = {"k_integrate": 16.0 * sum_w / n_launch_pass,                        # 8 B read + 8 B write per reference voxel update
                  "k_prepare": FRAME_BYTES_FIXED * frames_per_launch}                 # ScaleDepth: 2 B raw read + 4 B scaled write/read per pixel
            if warp_on:
                kb["k_reproject_scatter"] = 2.0 * FRAME_BYTES_RAW * frames_per_launch  # Reproject: 2 B read + 2 B scatter per pixel
            # (the committed rocprofv3 averages are those of configs[1]'s 50-frame launches: they price no other workload)
            by_kernel = dict(pj.get("rocprof_kernel_trace_avg_us_by_kernel") or {}) if static and args.config == 2 else {}
            if static and args.config == 2 and pj.get("rocprof_kernel_trace_avg_us") and "k_integrate" not in by_kernel:
                by_kernel = dict(by_kernel, k_integrate=float(pj["rocprof_kernel_trace_avg_us"]))
            kernels = {}
            for kname, nbytes in kb.items():
                e = {"algorithmic_bytes_per_launch": nbytes}
                if kname == "k_integrate":
                    e.update({"avg_launch_ms": ms_launch, "achieved": nbytes / (ms_launch * 1e-3) / 1e9,
                              "frac": nbytes / (ms_launch * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_source": "HIP events on the launch stream, this run"})
                if by_kernel.get(kname):
                    e.update({"rocprof_avg_us": float(by_kernel[kname]),
                              "frac_rocprof": nbytes / (float(by_kernel[kname]) * 1e-6) / 1e9 / HBM_PEAK_GBS})
                kernels[kname] = e
            kint = kernels["k_integrate"]
            out["roofline"] = {"bound": "hbm", "contract_bound": "hbm", "limiter": "latency + VALU issue, and the longest work items (DESIGN.md 4, 'Path A, round 3')",
                               "kernel": "k_integrate", "kernels_of_the_job": "k_integrate + k_prepare + k_reproject_scatter: the three kernels of a batch run concurrently on three streams",
                               "achieved": job_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": job_ach / HBM_PEAK_GBS,
                               "frac_definition": "SURVEY.md 8d's contract: (16 x sum(weight_) + 1 843 200 x F + 1 228 800 x F with the warp) / t_total / 8e12 -- "
                                                  "every algorithmic byte of the pass over the timed wall time of the pass (value x bytes per frame); recompute: "
                                                  "algorithmic_bytes_per_pass / (ms_per_step x steps x 1e-3) / 8e12.  Rounds 1-4 printed the k_integrate-only figure "
                                                  "here WITH the pre-pass bytes in its numerator (0.84 in round 4); that figure is gone, kernel_frac replaces it",
                               "algorithmic_bytes_per_pass": bytes_pass, "frames_per_pass": n_frames, "pass_ms": 1e3 * dt,
                               "whole_job_frac": job_ach / HBM_PEAK_GBS,
                               "kernel_frac": kint["frac"], "kernel_frac_rocprof": kint.get("frac_rocprof"),
                               "kernel_frac_definition": "k_integrate alone with ONLY its own bytes, 16 x (voxel updates per launch), over its average launch duration: "
                                                         "kernel_frac by HIP events inside the timed pipeline, kernel_frac_rocprof by the committed rocprofv3 "
                                                         "--kernel-trace --stats average (static_figures; tracing perturbs the three-stream overlap, so it reads lower)",
                               "kernels": kernels,
                               "frac_rocprof": kint.get("frac_rocprof"),
                               "static_figures": static,
                               "frac_of_measured_copy_peak": job_ach / HBM_COPY_GBS,
                               "measured_copy_peak": HBM_COPY_GBS, "traffic": traffic,
                               "traffic_source": ("static: %s, committed as profiles/pmc_latest.json (ONE run of k_integrate, 50-frame launch: "
                                                  "2 x FETCH_SIZE + WRITE_SIZE, an upper bound), per launch, not this run" % pj.get("run", "a rocprofv3 --pmc run"))
                               if traffic else None,
                               "hbm_physical_frac": phys,
                               "algorithmic_bytes_per_launch": per_launch, "avg_launch_ms": ms_launch, "launches": launches,
                               "launches_timed": timed_launches, "event_stride": args.event_stride,
                               "frames_per_launch": frames_per_launch,
                               "voxel_updates_per_pass": sum_w, "unit_visits": prof["unit_visits"],
                               "note": ("rank 0 kernel; voxel updates = job total / ranks; " if world > 1 else "") +
                                       "frac prices the ALGORITHMIC bytes of SURVEY.md 8d (16 B per reference voxel update) against the HBM "
                                       "peak, as the contract asks; the batched kernel moves about half of them (traffic, hbm_physical_frac); what it "
                                       "waits for is latency and the longest items of a launch (limiter), valu_issue.frac is its share of the VALU issue peak",
                               "valu_issue": valu}
            if alone and alone["launches"] > 0 and alone["integrate_ms"] > 0:
                ms_alone = alone["integrate_ms"] / alone["launches"]
                per_alone = 16.0 * sum_w / alone["launches"]                     # the kernel's own bytes only (rounds 1-4: the job's)
                out["roofline"]["kernel_alone"] = {
                    "avg_launch_ms": ms_alone, "launches": alone["launches"], "achieved": per_alone / (ms_alone * 1e-3) / 1e9,
                    "frac": per_alone / (ms_alone * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "what": "the same kernel and bytes in one extra untimed pass that synchronises after every 50-frame launch: "
                            "k_integrate alone on the chip, priced with 16 x (voxel updates per launch) only.  kernel_frac above is the same "
                            "figure inside the TIMED region, where the kernel shares the SIMDs with the pre-pass kernels of the next two "
                            "batches (three-stream pipeline)"}
        else:
            out["roofline"] = {"bound": "hbm", "contract_bound": "hbm", "kernel": "k_integrate", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": None, "traffic": None, "avg_launch_ms": ms_launch, "launches": launches}
        if streamed is not None:
            out["streamed"] = streamed
        if world == 1 and args.cpu_sample > 0 and warp_on and not dry and args.config != 2:
            out["cpu_baseline"], par = sampled_parity(sc, depth, args.cpu_sample, I, max_units, local)
            if par is not None:
                out["parity_checked"] = par
        elif world == 1 and args.cpu_sample > 0 and warp_on and not dry:
            ns = min(n_frames, max(I, (args.cpu_sample // I) * I))
            host = synth.to_numpy_u16(depth[:ns])
            with tempfile.TemporaryDirectory() as fdir:
                out["cpu_baseline"], ref_units = cpu_baseline(sc, host, ns, fdir)
                # the same ns frames once more on the GPU -- through the host mirror of CIntegrateApp reading the SAME pose.log /
                # seg.log / g.ctr the reference just read -- compared unit by unit with the volume the CPU run left behind
                from elasticreconstruction_amd.tsdf import IntegrateApp
                app = IntegrateApp(max_units=max_units, device=local)
                if out["cpu_baseline"]["kind"] == "reference":
                    app.pose_filename_, app.seg_filename_, app.ctr_filename_ = (os.path.join(fdir, n) for n in ("pose.log", "seg.log", "g.ctr"))
                    app.ctr_num_, app.ctr_resolution_, app.ctr_length_, app.ctr_interval_ = ns // I, sc["resolution"], sc["length"], I
                    app.Init()
                    for f in range(ns):
                        app.Execute(f + 1, host[f])
                    app.Finish(save=False)
                    pvol = app.volume_
                else:                                        # oracle port: it was fed the in-memory matrices
                    vol.reset()
                    for lo in range(0, ns, I):
                        vol.IntegrateFrames(None, sc["traj"][lo:lo + I], warp_slice(lo, lo + I), device_ptr=depth.data_ptr() + lo * px * 2)
                    vol.synchronize()
                    pvol = vol
                out["parity_checked"] = parity_check(pvol, ref_units, ns, out["cpu_baseline"]["kind"])
                if pvol is not vol:
                    pvol.close()
        if icp is not None:
            out["icp"] = icp
            if world == 1 and args.config == 2 and not dry:
                out["fragment_optimizer"] = fopt_section(local)
        if world > 1 and args.config == 2 and args.scaling_child and strong is not None:
            out["other_configs"] = {"configs[3]": strong}
        if world == 1 and args.config == 2 and args.boundary and warp_on and n_frames == CONFIG2_FRAMES and not dry:
            vol.close()                                              # (the programs bring their own volumes)
            vol = None
            out["boundary"] = boundary_section(sc, depth, local)
        if world == 1 and args.config == 2 and args.other_configs and not args.host_input and not args.force_merge and not dry:
            if vol is not None:
                vol.close()
            vol = None
            out["other_configs"] = other_configs(local)
        emit(out, args.full_json)
    if vol is not None:
        vol.close()
    if comm is not None:
        comm.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
