"""The N > 1 paths on CPU: world_size 2 over gloo.  The frame-split merge protocol (key all-gather + ONE
all-reduce of the sdf*weight / weight planes, elasticreconstruction_amd/parallel.py) is the same code
bench.py runs over RCCL; here it is driven with a host-memory volume filled by the oracle so that it needs
no GPU.  Pair sharding has no collective on the data path; only the host gather is exercised."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VOX = 64 ** 3


class HostVolume:
    """unit_keys / export_weighted / import_weighted / synchronize over numpy, filled by the CPU oracle."""

    def __init__(self):
        self.units = {}

    def integrate_with_oracle(self, depth, poses):
        from oracle.pyoracle import OracleVolume
        ora = OracleVolume()
        for d, T in zip(depth, poses):
            ora.Integrate(d, T)
        for k in ora.unit_keys():
            self.units[int(k)] = ora.read_unit(k)

    def unit_keys(self):
        return np.array(sorted(self.units), np.int32)

    @staticmethod
    def _view(ptr, n):
        return np.ctypeslib.as_array((C.c_float * (n * 2 * VOX)).from_address(ptr)).reshape(n, 2, VOX)

    def export_weighted(self, keys, ptr):
        buf = self._view(ptr, len(keys))
        for q, k in enumerate(keys):
            if int(k) in self.units:
                s, w = self.units[int(k)]
                buf[q, 0], buf[q, 1] = s * w, w
            else:
                buf[q] = 0

    def import_weighted(self, keys, ptr):
        buf = self._view(ptr, len(keys))
        for q, k in enumerate(keys):
            sw, w = buf[q, 0].copy(), buf[q, 1].copy()
            with np.errstate(divide="ignore", invalid="ignore"):
                s = np.where(w > 0, sw / w, np.float32(0)).astype(np.float32)
            self.units[int(k)] = (s, w)

    def export_raw(self, keys, ptr):
        buf = self._view(ptr, len(keys))
        for q, k in enumerate(keys):
            buf[q, 0], buf[q, 1] = self.units[int(k)]               # (only asked for units this rank owns)

    def import_raw(self, keys, ptr):
        buf = self._view(ptr, len(keys))
        for q, k in enumerate(keys):
            self.units[int(k)] = (buf[q, 0].copy(), buf[q, 1].copy())

    def synchronize(self):
        pass


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["ER_ORACLE_QUIET"] = "1"
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    from elasticreconstruction_amd import parallel, synth
    n = 6
    poses = synth.circle_trajectory(3000, revolutions=1.0)[::40][:n]
    depth = synth.to_numpy_u16(synth.render_depth(poses))
    lo, hi = parallel.frame_block(n, rank, world)
    vol = HostVolume()
    vol.integrate_with_oracle(depth[lo:hi], poses[lo:hi])
    local_keys = vol.unit_keys()
    import copy
    vol_ar = copy.deepcopy(vol)
    n_union = parallel.merge_volumes(vol, dist, torch.device("cpu"))                       # reduce to rank 0
    n_union_ar = parallel.merge_volumes(vol_ar, dist, torch.device("cpu"), mode="all_reduce")
    same = n_union == n_union_ar and all(np.array_equal(vol_ar.units[k][1], vol.units[k][1]) for k in vol.units) if rank == 0 else True
    # a rank-local failure must become a COLLECTIVE one: rank 1's key query raises like an overflowed unit pool;
    # rank 0 must come back with MergeError instead of blocking in the reduction, rank 1 with its own error
    class Broken(HostVolume):
        def unit_keys(self):
            raise RuntimeError("TSDF unit pool exhausted (test)")
    probe = Broken() if rank == 1 else copy.deepcopy(vol)
    try:
        parallel.merge_volumes(probe, dist, torch.device("cpu"))
        kind = "none"
    except parallel.MergeError:
        kind = "peer"
    except RuntimeError:
        kind = "local"
    kinds = [None] * world
    dist.all_gather_object(kinds, kind)
    # pair sharding: every pair exactly once, results back in order
    mine = {p: (p, rank) for p in parallel.pair_shard(7, rank, world)}
    gathered = parallel.gather_pair_results(mine, 7, dist)
    keysets = [None] * world
    dist.all_gather_object(keysets, [int(k) for k in local_keys])
    if rank == 0:
        full = HostVolume()
        full.integrate_with_oracle(depth, poses)
        ok_keys = np.array_equal(vol.unit_keys(), full.unit_keys())
        worst, wbad = 0.0, 0
        for k in full.unit_keys():
            s, w = vol.units[int(k)]
            sf, wf = full.units[int(k)]
            wbad += int((w != wf).sum())
            worst = max(worst, float(np.abs(s - sf).max()))
        # round 5, the sparse merge: a unit only ONE rank touched travels raw and must equal the single-volume result BIT FOR BIT (no frame of the
        # other rank ever reached it), in the reduce AND in the all-reduce mode
        single = sorted(set(keysets[0]) ^ set(keysets[1]))
        raw_exact = all(np.array_equal(v.units[k][0].view(np.uint32), full.units[k][0].view(np.uint32)) and np.array_equal(v.units[k][1], full.units[k][1])
                        for v in (vol, vol_ar) for k in single)
        q.put(dict(failure_kinds=kinds, ok_keys=ok_keys, worst=worst, wbad=wbad, n_union=n_union, n_local=len(local_keys), modes_agree=bool(same),
                   single_units=len(single), multi_units=len(set(keysets[0]) & set(keysets[1])), raw_exact=bool(raw_exact),
                   pairs=[g[0] for g in gathered], owners=[g[1] for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_split_merge_and_pair_shard_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["failure_kinds"] == ["peer", "local"], res["failure_kinds"]
    assert res["ok_keys"], "union of the per-rank unit keys != single-volume keys"
    assert res["wbad"] == 0, "merged weights must be exact (integer-valued floats)"
    assert res["worst"] <= 1e-5, "merged tsdf off by %.3g" % res["worst"]
    assert res["n_union"] >= res["n_local"] > 0 and res["modes_agree"]
    assert res["single_units"] > 0 and res["multi_units"] > 0 and res["raw_exact"], (res["single_units"], res["multi_units"], res["raw_exact"])
    assert res["pairs"] == list(range(7)) and res["owners"] == [0, 1, 0, 1, 0, 1, 0]


def test_frame_block_partition():
    from elasticreconstruction_amd import parallel
    for n, w in ((10000, 8), (3000, 4), (7, 3), (2, 4)):
        blocks = [parallel.frame_block(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))


def test_c_abi_shard_helpers_match_the_python_protocol():
    """The pieces of the multi-GPU host protocol that live behind the C ABI and need no GPU: er_frame_block (the contiguous
    frame blocks bin/Integrate --gpus cuts) equals parallel.frame_block, and er_unit_owner (--shard unit) assigns every unit
    of a 512^3-unit lattice to exactly one of `world` ranks, evenly along every lattice line."""
    import ctypes as C
    from elasticreconstruction_amd import _ffi, parallel
    L = _ffi.lib()
    lo, hi = C.c_int(), C.c_int()
    for n, w in ((10000, 8), (3000, 4), (3000, 7), (7, 3), (2, 4), (0, 2), (5, 1)):
        for r in range(w):
            L.er_frame_block(n, r, w, C.byref(lo), C.byref(hi))
            assert (lo.value, hi.value) == parallel.frame_block(n, r, w)
    rng = np.random.default_rng(3)
    for world in (1, 2, 3, 8):
        xyz = rng.integers(200, 312, (2000, 3))
        keys = xyz[:, 0] * 512 * 512 + xyz[:, 1] * 512 + xyz[:, 2]
        owners = np.array([L.er_unit_owner(int(k), world) for k in keys])
        assert ((owners >= 0) & (owners < world)).all() and np.array_equal(owners, xyz.sum(1) % world)
        # the 8x8x8 units of the 512^3 region: every rank gets its share (+-1 lattice plane)
        region = np.array([[x, y, z] for x in range(252, 260) for y in range(252, 260) for z in range(252, 260)])
        cnt = np.bincount(region.sum(1) % world, minlength=world)
        assert cnt.max() - cnt.min() <= 64


def test_c_merge_protocol_world_1_2_3_on_threads(tmp_path):
    """The protocol of er_tsdf_allreduce (csrc/er_merge_protocol.h -- the very header er_multi.hip runs over RCCL) on host
    threads with a shared-memory transport, world = 1 / 2 / 3: uneven key counts, empty ranks, root 0 / 1 / 2 / all-reduce, and
    the collective-failure rule (a rank whose unit pool overflowed or whose export failed makes EVERY rank return an error after
    the same collective -- no rank is left waiting; the timeout is the hang detector).  Also under ThreadSanitizer when g++ has it."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "tests", "cpp", "merge_protocol_check.cpp")
    inc = os.path.join(root, "elasticreconstruction_amd", "csrc")
    exe = str(tmp_path / "merge_protocol_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-Wall", "-Werror", "-I" + inc, src, "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.startswith("OK "), r.stdout + r.stderr
    tsan = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", "-I" + inc, src, "-o", exe + "_tsan"],
                          capture_output=True, text=True)
    if tsan.returncode == 0:
        r = subprocess.run([exe + "_tsan"], capture_output=True, text=True, timeout=180)
        assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr, r.stdout + r.stderr


def json_line(stdout):
    import json
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 must print exactly ONE JSON line, got %d:\n%s" % (len(lines), stdout[-2000:])
    return json.loads(lines[0])


def _bench_dry_run(nproc, extra, timeout=420):
    """`python -m torch.distributed.run ... bench.py --gpus N ... --dry-run` exactly as the driver launches the real thing (README /
    the task contract), on CPU: returns the parsed JSON line of rank 0."""
    import json
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import tempfile
    full = os.path.join(tempfile.mkdtemp(prefix="er_bench_dry_"), "bench_full.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--dry-run", "--full-json", full] + extra
    env = dict(os.environ, OMP_NUM_THREADS="1", ER_ORACLE_QUIET="1")
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    compact = json_line(r.stdout)
    check_compact_line(r.stdout, compact)
    with open(full) as fh:
        out = json.load(fh)
    # the compact line is made FROM the full object: the headline numbers agree
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "scaling", "dtype", "data", "higher_is_better", "vs_baseline"):
        assert compact[k] == out[k], k
    assert abs(compact["value"] - out["value"]) <= 1e-6 * out["value"] and abs(compact["ms_per_step"] - out["ms_per_step"]) <= 1e-6 * out["ms_per_step"]
    return out


COMPACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline"}


def check_compact_line(stdout, c):
    """BENCH_r05.json parsed as null because the one stdout line had grown to 22 KB.  The LAST stdout line must be a JSON
    object under 4 KB with the contract's keys, and roofline.frac must recompute from the line itself."""
    import json
    last = [l for l in stdout.splitlines() if l.strip()][-1]
    assert len(last) < 4096, "compact line is %d bytes" % len(last)
    assert json.loads(last) == c
    assert COMPACT_KEYS <= set(c), COMPACT_KEYS - set(c)
    assert {"workload", "baseline_config", "frames_per_step", "volume_units_touched", "inputs"} <= set(c["config"])
    assert c["config"]["workload"].startswith("configs[") and len(c["config"]["workload"]) < 100
    rf = c["roofline"]
    assert {"bound", "kernel", "frac", "achieved", "peak", "unit", "kernel_frac", "algorithmic_bytes_per_pass", "traffic", "avg_launch_ms"} <= set(rf)
    assert rf["bound"] == "hbm" and rf["kernel"] == "k_integrate" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    frac = rf["algorithmic_bytes_per_pass"] / (c["ms_per_step"] * c["steps"] * 1e-3) / 8e12
    assert abs(rf["frac"] - frac) <= 1e-5 * frac, (rf["frac"], frac)
    assert abs(rf["achieved"] / rf["peak"] - rf["frac"]) <= 1e-5 * frac
    # k_integrate priced with its own bytes over its own launch time
    assert abs(rf["kernel_frac"] - rf["kernel_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 8e12) <= 1e-5 * rf["kernel_frac"]

    def no_prose(x, path="line"):
        if isinstance(x, dict):
            for k, v in x.items():
                no_prose(v, path + "." + k)
        elif isinstance(x, str):
            assert len(x) <= 100, "%s carries %d characters of text" % (path, len(x))
    no_prose(c)


def test_bench_dry_run_four_ranks_through_the_drivers_command_line():
    """nothing with world > 1 had ever run through bench.py's own control flow.  This launches the driver's
    exact command line with 4 ranks and --dry-run (gloo, host-array volume, the merge protocol of parallel.merge_volumes behind
    the stand-in communicator) and checks the JSON contract: one line from rank 0, whole-job value, weak scaling, the merge's
    union (one unit shared by all ranks + one private unit per rank and launch), rccl_ranks, the sharded ICP figure, and that the
    summed unit weights of the merged volume are those of every rank's frames (unit weights add exactly)."""
    out = _bench_dry_run(4, ["--steps", "2", "--warmup", "1", "--frames-per-step", "50", "--icp-pairs", "5", "--min-seconds", "0.05"])
    assert out["dry_run"] is True and out["n_gpus"] == 4 and out["steps"] == 2 and out["warmup"] == 1
    assert out["metric"].startswith("depth frames/sec") and out["unit"] == "frames/s" and out["higher_is_better"] is True
    assert out["scaling"] == "weak" and out["vs_baseline"] is None and out["dtype"] == "f32" and out["data"] == "synthetic"
    assert out["value"] > 0 and abs(out["value"] * out["ms_per_step"] * 1e-3 * out["steps"] - 4 * 100) < 1e-6 * 400      # value = ALL ranks' frames / time
    cfg = out["config"]
    assert cfg["rccl_ranks"] == 4 and cfg["merge_impl"].startswith("abi") and cfg["frames_per_gpu"] == 100
    assert cfg["merge_union_units"] == 1 + 4                          # the shared unit + one private unit per rank (100 frames: (frames // 64) % 8 = 0, 0)
    assert "frame-block shard x4" in cfg["parallelism"]
    assert out["roofline"]["kernel"] == "k_integrate" and out["roofline"]["launches_timed"] > 0
    # the contract figure can be recomputed from the line itself: every algorithmic byte of the pass over the timed wall time of the pass
    rf = out["roofline"]
    assert abs(rf["frac"] - rf["algorithmic_bytes_per_pass"] / (out["ms_per_step"] * out["steps"] * 1e-3) / 8e12) < 1e-9 * max(rf["frac"], 1.0)
    assert rf["frac"] == rf["whole_job_frac"] and set(rf["kernels"]) == {"k_integrate", "k_prepare", "k_reproject_scatter"}
    assert abs(rf["kernels"]["k_integrate"]["algorithmic_bytes_per_launch"] - 16.0 * rf["voxel_updates_per_pass"] / (rf["launches"] / out["timing"]["passes"])) < 1.0
    # after the reduce rank 0 holds every rank's updates: sum(weight) / world = one rank's share = 2 units x 100 frames x 64^3 voxels
    assert out["roofline"]["voxel_updates_per_pass"] == 2 * 100 * 64 ** 3
    assert out["icp"]["pairs"] == 20 and "4 GPUs x 5 pairs" in out["icp"]["sharding"]
    assert "cpu_baseline" not in out
    # N > 1: configs[3]'s strong-scaling job ran over the same communicator: 10 000 frames in 4 blocks of 2500, whole-job value
    sc3 = out["other_configs"]["configs[3]"]
    assert sc3["scaling"] == "strong" and sc3["rccl_ranks"] == 4 and sc3["frames"] == 10000 and sc3["frames_per_gpu"] == 2500
    assert abs(sc3["value"] * sc3["ms_per_step"] * 1e-3 * sc3["steps"] - 10000) < 1e-6 * 10000


def test_compact_line_of_a_full_headline_object_stays_under_4k():
    """The compact line from the LARGEST object bench.py has ever produced (profiles/r05o_bench_default.json, 22 KB: the line the driver could
    not parse in round 5) is under 4 KB, keeps the headline, roofline, cpu_baseline, parity, ICP and other-config figures, and drops the prose."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "r05o_bench_default.json")) as fh:
        out = json.loads([l for l in fh.read().splitlines() if l.startswith("{")][-1])
    line = bench.compact_line(out)
    assert len(line) < 4096, len(line)
    c = json.loads(line)
    check_compact_line(line, c)
    assert abs(c["value"] - out["value"]) < 1e-6 * out["value"] and c["config"]["workload"] == "configs[1]" and c["config"]["inputs"] == "hbm_resident"
    assert c["cpu_baseline"]["kind"] == "reference" and c["cpu_baseline"]["cores"] == 8 and c["cpu_baseline"]["value"] > 0
    assert c["parity_checked"]["bit_exact"] is True
    assert c["icp"]["pairs_per_s"] > 0 and c["icp"]["realistic_pairs_per_s"] > 0 and c["icp"]["cpu_baseline"]["kind"] == "reference"
    assert set(c["other_configs"]) == {"configs[3]", "configs[4]"} and c["other_configs"]["configs[3]"]["bit_exact"] is True


def test_bench_dry_run_config4_and_config5_two_ranks():
    """configs[3] (strong scaling: the JOB's 10 000 frames are split over the ranks; K is derived from the world size) and configs[4]
    (the all-pairs figure reduced over the ranks, then 5000 frames) through the same launcher with 2 ranks."""
    out = _bench_dry_run(2, ["--config", "4", "--min-seconds", "0.01", "--max-passes", "2"])
    assert out["scaling"] == "strong" and out["n_gpus"] == 2 and out["config"]["frames_per_gpu"] == 5000 and out["steps"] == 25
    assert out["config"]["frames_per_step"] == 200
    assert out["config"]["rccl_ranks"] == 2 and out["config"]["merge_union_units"] == 1 + 2 * 8 and out["config"]["baseline_config"] == 4
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 * out["steps"] - 10000) < 1e-6 * 10000
    out = _bench_dry_run(2, ["--config", "5", "--min-seconds", "0.01", "--max-passes", "2"])
    assert out["scaling"] == "strong" and out["config"]["frames_per_gpu"] == 2500 and out["config"]["baseline_config"] == 5
    assert out["icp"]["pairs_total"] == 4950 and out["icp"]["pairs_per_s"] > 0


def test_bench_step_plan_covers_the_jobs_frames_at_1_2_4_8_ranks():
    """bench.plan_steps: configs[3]'s 10 000 frames are met exactly at 1 / 2 / 4 / 8 ranks, configs[4]'s 5000 exactly up to 4 ranks and
    rounded UP to whole fragments at 8 (650 per rank); the headline's 20 steps cover configs[1]'s 3000 frames."""
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 4, 8):
        K, S = bench.plan_steps(4, 20, 0, 50, world)
        assert K * S * world == 10000 and S % 50 == 0, (world, K, S)
        K, S = bench.plan_steps(5, 20, 0, 50, world)
        assert K * S == {1: 5000, 2: 2500, 4: 1250, 8: 650}[world] and S % 50 == 0
        assert bench.plan_steps(2, 20, 0, 50, world) == (20, 150)
    assert bench.plan_steps(2, 5, 0, 50, 1) == (5, 600) and bench.plan_steps(2, 7, 100, 50, 1) == (7, 100)


def test_compact_line_sheds_appendices_before_it_loses_the_headline():
    """Whatever the optional legs put into the full object, the stdout line stays under 4 KB and keeps metric / value / roofline / cpu_baseline: the
    appendices (fragment_optimizer, boundary, streamed, other_configs, icp) are dropped, largest first, if they ever push it over."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    with open(os.path.join(ROOT, "profiles", "r05o_bench_default.json")) as fh:
        out = json.loads([l for l in fh.read().splitlines() if l.startswith("{")][-1])
    out["other_configs"] = {"configs[%d]" % k: dict(out["other_configs"]["configs[3]"], workload="x" * 50) for k in range(3, 60)}   # 57 children
    line = bench.compact_line(out)
    assert len(line) < 4096
    c = json.loads(line)
    assert "other_configs" not in c and c["value"] > 0 and c["roofline"]["frac"] > 0 and c["cpu_baseline"]["kind"] == "reference"
    # numbers only carry 8 significant digits, strings stay short, NaN / inf never reach the line
    out2 = dict(out, other_configs=None)
    out2["roofline"] = dict(out["roofline"], frac=float("nan"), achieved=float("inf"))
    c2 = json.loads(bench.compact_line(out2))
    assert c2["roofline"]["frac"] is None and c2["roofline"]["achieved"] is None
