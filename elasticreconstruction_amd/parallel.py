"""Multi-GPU sharding of the two paths: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

  Path B (pairs)   fully independent units: pair p -> rank p mod G, fragments replicated on every GPU,
                   results gathered on rank 0 in the original pair order.  NO data-path collective.
  Path A (frames)  contiguous frame blocks per rank into private volumes, then ONE exchange step:
                   all-gather of the touched unit keys -> union, ONE reduce(sum) to rank 0 (or all-reduce) over
                   the [key][sdf*weight | weight] planes of the union, per-voxel divide on import.
                   (Since round 5 only the units two or more ranks touched go through the sum; a unit one rank touched travels raw,
                   bit for bit, from its owner to where the result is wanted: csrc/er_merge_protocol.h -- er_tsdf_allreduce behind
                   AbiComm is the product path, merge_volumes below the torch.distributed cross-check of the same protocol.)
                   (The running mean with unit weights is a sum: w = sum_g w_g, sdf = sum_g sdf_g*w_g / w;
                   TSDFVolume.cpp:93-94 applied sequentially gives the same value up to float rounding
                   order, hence tolerance 1e-5 instead of bit parity for this mode.)

The functions only need an object with unit_keys() / export_weighted(keys, ptr) / import_weighted(keys, ptr) / export_raw(keys, ptr) /
import_raw(keys, ptr) / synchronize(), so the CPU tests can drive the identical protocol over gloo with a host-memory volume.
"""
import numpy as np


def frame_block(n_frames, rank, world):
    """Contiguous block [lo, hi) of rank `rank` (config 4: frame-batch shard)."""
    per = (n_frames + world - 1) // world
    lo = min(rank * per, n_frames)
    return lo, min(lo + per, n_frames)


def pair_shard(n_pairs, rank, world):
    """Static cyclic assignment of fragment pairs (the reference uses OpenMP schedule(dynamic), CorresApp.cpp:121,220)."""
    return list(range(rank, n_pairs, world))


class MergeError(RuntimeError):
    """The frame-split merge was abandoned by ALL ranks together because one of them failed locally."""


def _agree_max(values, dist, device):
    """all_reduce(MAX) of a few host ints (control plane of the merge)."""
    import torch
    t = torch.tensor(list(values), dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [int(v) for v in t.cpu()]


def gather_keys(local_keys, dist, device, status=0):
    """Every rank's touched unit keys on every rank: the ranks first AGREE on the padded length -- one all_reduce(MAX) of {key count, status} --
    then exchange the keys in ONE fixed-size all-gather (padded with -1), so every rank issues the same collectives with the same shapes
    whatever it touched (csrc/er_merge_protocol.h, steps 2-3).  status != 0 on any rank makes every rank raise MergeError after the first
    collective.  Returns [sorted int32 numpy array of rank q's keys for q in range(world)]."""
    import torch
    world = dist.get_world_size()
    keys = np.ascontiguousarray(local_keys, np.int32)
    max_keys, any_failed = _agree_max([keys.size, 1 if status else 0], dist, device)
    if any_failed:
        raise MergeError("a rank failed before the merge (its unit pool or hash table overflowed?); nothing was merged")
    if max_keys <= 0:
        return [np.zeros(0, np.int32) for _ in range(world)]
    pad = torch.full((max_keys,), -1, dtype=torch.int32)
    if keys.size:
        pad[:keys.size] = torch.from_numpy(keys)
    pad = pad.to(device)
    allk = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(allk, pad)                               # (list form: also supported by gloo in the CPU tests)
    out = []
    for t in allk:
        a = np.unique(t.cpu().numpy())
        out.append(a[a >= 0].astype(np.int32))
    return out


def union_keys(local_keys, dist, device, status=0):
    """Sorted union of the per-rank touched unit keys (gather_keys, then the union)."""
    per_rank = gather_keys(local_keys, dist, device, status)
    return np.unique(np.concatenate(per_rank)).astype(np.int32) if per_rank else np.zeros(0, np.int32)


def merge_plan(per_rank_keys, rank, root):
    """Who touched what (csrc/er_merge_protocol.h step 3, the same arithmetic): (union, multi, send, recv) with multi = the keys two or more ranks
    touched (they go through the sum), send = this rank's single-toucher keys that have to travel (to `root`, or to everybody for root < 0),
    recv = {owner: keys} of the single-toucher units that arrive here.  All arrays sorted."""
    world = len(per_rank_keys)
    allk = np.concatenate(per_rank_keys) if world else np.zeros(0, np.int32)
    union, cnt = np.unique(allk, return_counts=True)
    multi = union[cnt >= 2].astype(np.int32)
    single = set(int(k) for k in union[cnt == 1])
    send, recv = np.zeros(0, np.int32), {}
    for q in range(world):
        own = np.array(sorted(int(k) for k in per_rank_keys[q] if int(k) in single), np.int32)
        travels = own.size > 0 and (world > 1 if root < 0 else q != root)
        if not travels:
            continue
        if q == rank:
            send = own
        elif root < 0 or root == rank:
            recv[q] = own
    return union.astype(np.int32), multi, send, recv


def merge_volumes(vol, dist, device, sync_stream=None, mode="reduce", root=0):
    """Frame-split merge over torch.distributed -- the cross-check of the product path (er_tsdf_allreduce, same protocol:
    csrc/er_merge_protocol.h, since round 5 including its sparse data path).  mode "reduce" (default): ONE reduce(sum) to `root` over the
    [key][sdf*weight | weight] planes of the units TWO OR MORE ranks touched -- the "final reduce of per-GPU TSDF volume-unit weights" of
    BASELINE.json -- and one send per owner of the units only ONE rank touched, raw, bit for bit; afterwards `root` holds the complete volume
    (the other ranks keep their partial volumes).  mode "all_reduce": an all-reduce and one broadcast per owner; every rank ends with the
    complete volume.  Returns the union size.
    A rank-local failure (vol.unit_keys() / export raising, e.g. "raise max_units") is carried through the next collective
    as a status: every rank raises MergeError together instead of the healthy ones blocking in the reduction.
    sync_stream: callable that orders the communication stream after the volume's kernels and vice versa.  None (default)
    is SAFE for any stream set-up: the volume's streams are drained after the export and torch's current stream (the one
    the collective is ordered on) is drained before the import -- two host waits, once per job."""
    import torch
    local_error = None
    rank, world = dist.get_rank(), dist.get_world_size()
    all_mode = mode == "all_reduce"
    try:
        keys = vol.unit_keys()
    except Exception as ex:                                   # unit pool / hash table overflow on THIS rank
        keys, local_error = np.zeros(0, np.int32), ex
    try:
        per_rank = gather_keys(keys, dist, device, status=1 if local_error else 0)
    except MergeError:
        if local_error:
            raise local_error
        raise
    union, multi, send, recv = merge_plan(per_rank, rank, -1 if all_mode else root)
    if union.size == 0:
        return 0
    plane = 2 * 64 ** 3

    def wait_volume():
        if sync_stream:
            sync_stream()
        else:
            vol.synchronize()

    def wait_comm():
        if sync_stream:
            sync_stream()
        elif getattr(device, "type", str(device)) != "cpu" and str(device) != "cpu":
            torch.cuda.current_stream(device).synchronize()

    buf = sbuf = None
    rbuf = {}
    try:
        if multi.size:
            buf = torch.empty((multi.size, 2, 64 ** 3), dtype=torch.float32, device=device)
            vol.export_weighted(multi, buf.data_ptr())
        if send.size:
            sbuf = torch.empty((send.size, 2, 64 ** 3), dtype=torch.float32, device=device)
            vol.export_raw(send, sbuf.data_ptr())
        for q, ks in recv.items():
            rbuf[q] = torch.empty((ks.size, 2, 64 ** 3), dtype=torch.float32, device=device)
        wait_volume()                                        # the export kernels have written buf / sbuf
    except Exception as ex:
        local_error = ex
    if _agree_max([1 if local_error else 0], dist, device)[0]:
        if local_error:
            raise local_error
        raise MergeError("a rank failed while exporting its planes; nothing was merged")
    if multi.size:                                           # (the same decision on every rank: `multi` is a function of the gathered keys)
        if all_mode:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)       # the ONLY arithmetic collective of the pipeline
        else:
            dist.reduce(buf, dst=root, op=dist.ReduceOp.SUM)
    # the single-toucher units: owners in ascending rank order, so that every rank issues matching calls in the same order
    single_cnt = [int(np.setdiff1d(per_rank[q], multi).size) for q in range(world)]
    for q in range(world):
        if single_cnt[q] == 0:
            continue
        if all_mode:
            if world > 1:
                t = sbuf if q == rank else rbuf[q]
                dist.broadcast(t, src=q)
        elif q != root:
            if q == rank:
                dist.send(sbuf, dst=root)
            elif rank == root:
                dist.recv(rbuf[q], src=q)
    wait_comm()                                              # the reduce / the received units have landed
    if all_mode or rank == root:
        if multi.size:
            vol.import_weighted(multi, buf.data_ptr())
        for q, ks in recv.items():
            vol.import_raw(ks, rbuf[q].data_ptr())
    vol.synchronize()
    del plane
    return int(union.size)


def gather_pair_results(local_results, n_pairs, dist):
    """Host gather of per-pair results (dicts keyed by pair index) onto every rank, original order."""
    world = dist.get_world_size()
    allr = [None] * world
    dist.all_gather_object(allr, local_results)
    out = [None] * n_pairs
    for part in allr:
        for k, v in part.items():
            out[k] = v
    return out


MERGE_ALL, MERGE_DISTRIBUTED = -1, -2        # er_hip.h: ER_MERGE_ALL, ER_MERGE_DISTRIBUTED


def _merge_stats(lib, handle):
    """er_comm_merge_stats + er_comm_merge_stats_owner of one communicator handle as a dict."""
    import ctypes as C
    from . import _ffi
    st = (C.c_longlong * 8)()
    _ffi.check(lib.er_comm_merge_stats(handle, st), "er_comm_merge_stats")
    names = ("union_units", "multi_toucher_units", "single_toucher_units", "units_sent", "units_received", "bytes_reduced", "bytes_sent", "bytes_received")
    out = {n: int(v) for n, v in zip(names, st)}
    so = (C.c_longlong * 12)()
    _ffi.check(lib.er_comm_merge_stats_owner(handle, so), "er_comm_merge_stats_owner")
    out["impl"] = "owner" if so[0] == 1 else "ring"
    if so[0] == 1:
        out.update({"units_owned": int(so[4]), "units_summed_here": int(so[5]), "units_handed_over": int(so[6]),
                    "to_owners_bytes_sent": int(so[7]), "to_owners_bytes_received": int(so[8]),
                    "to_root_bytes_sent": int(so[9]), "to_root_bytes_received": int(so[10]), "ring_equivalent_bytes": int(so[11])})
    return out


class AbiComm:
    """The frame-split merge through liber_hip.so's OWN RCCL calls (er_comm_* / er_tsdf_allreduce, what bin/Integrate --gpus
    uses) for a one-process-per-GPU job: rank 0 draws the 128-byte communicator id, torch.distributed only carries it to the
    other ranks (out of band), every rank then creates its communicator and the collectives are issued from the library."""

    def __init__(self, dist, device_index):
        import ctypes as C
        from . import _ffi
        self._lib = _ffi.lib()
        rank, world = dist.get_rank(), dist.get_world_size()
        ident = (C.c_ubyte * 128)()
        if rank == 0:
            _ffi.check(self._lib.er_comm_unique_id(ident), "er_comm_unique_id")
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=0)
        ident = (C.c_ubyte * 128).from_buffer_copy(box[0])
        h = C.c_void_p()
        _ffi.check(self._lib.er_comm_create(ident, rank, world, int(device_index), C.byref(h)), "er_comm_create")
        self._h = h

    def allreduce(self, vol, root=0):
        """root >= 0: the merged volume on `root`; MERGE_ALL (-1): on every rank; MERGE_DISTRIBUTED (-2): every unit complete on its owner, nothing
        gathered (er_hip.h).  Returns the size of the key union."""
        import ctypes as C
        from . import _ffi
        n = C.c_int(0)
        _ffi.check(self._lib.er_tsdf_allreduce(vol._h, self._h, int(root), C.byref(n)), "er_tsdf_allreduce")
        return n.value

    def merge_stats(self):
        """What this rank's last allreduce moved (er_comm_merge_stats / er_comm_merge_stats_owner)."""
        return _merge_stats(self._lib, self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.er_comm_destroy(self._h)
            self._h = None


class LoopbackComms:
    """n ranks of ONE process on ONE device (er_comm_create_loopback): the product's merge protocol over real device volumes without RCCL -- the sum
    reduction is a kernel that adds the ranks' plane buffers in rank order, the point-to-point step device-to-device copies.  allreduce(vols, root) calls
    er_tsdf_allreduce for every rank from a thread of its own (ctypes releases the GIL), as `bin/Integrate --gpus N --same_device` does."""

    def __init__(self, n, device=0):
        import ctypes as C
        from . import _ffi
        self._lib = _ffi.lib()
        self.n = int(n)
        arr = (C.c_void_p * self.n)()
        _ffi.check(self._lib.er_comm_create_loopback(self.n, int(device), arr), "er_comm_create_loopback")
        self._h = [C.c_void_p(arr[i]) for i in range(self.n)]

    def allreduce(self, vols, root=0):
        """vols[r] = rank r's volume.  Returns the union size (the same on every rank)."""
        import ctypes as C
        from concurrent.futures import ThreadPoolExecutor
        from . import _ffi
        assert len(vols) == self.n

        def one(r):
            n = C.c_int(0)
            rc = self._lib.er_tsdf_allreduce(vols[r]._h, self._h[r], int(root), C.byref(n))
            return rc, n.value, (self._lib.er_last_error().decode() if rc else "")
        with ThreadPoolExecutor(self.n) as ex:
            out = list(ex.map(one, range(self.n)))
        bad = [o for o in out if o[0]]
        if bad:
            raise _ffi.ErError("er_tsdf_allreduce (loopback): " + bad[0][2])
        return out[0][1]

    def merge_stats(self, rank=0):
        return _merge_stats(self._lib, self._h[rank])

    def allreduce_failing(self, vols, root=0):
        """Like allreduce, but returns every rank's (rc, message) instead of raising: the collective-failure tests."""
        import ctypes as C
        from concurrent.futures import ThreadPoolExecutor

        def one(r):
            n = C.c_int(0)
            rc = self._lib.er_tsdf_allreduce(vols[r]._h, self._h[r], int(root), C.byref(n))
            return rc, (self._lib.er_last_error().decode() if rc else "")
        with ThreadPoolExecutor(self.n) as ex:
            return list(ex.map(one, range(self.n)))

    def close(self):
        for h in getattr(self, "_h", []):
            self._lib.er_comm_destroy(h)
        self._h = []


class LocalComms(LoopbackComms):
    """One process, n DISTINCT GPUs, one RCCL communicator each (er_comm_create_local = ncclCommInitAll) and one host thread per rank: what
    `bin/Integrate --gpus N` does, for tests on a box with several GPUs (tests/test_distributed_gpu.py).  Same interface as LoopbackComms."""

    def __init__(self, devices):
        import ctypes as C
        from . import _ffi
        self._lib = _ffi.lib()
        self.n = len(devices)
        dv = (C.c_int * self.n)(*[int(d) for d in devices])
        arr = (C.c_void_p * self.n)()
        _ffi.check(self._lib.er_comm_create_local(self.n, dv, arr), "er_comm_create_local")
        self._h = [C.c_void_p(arr[i]) for i in range(self.n)]
