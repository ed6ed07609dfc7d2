"""Multi-GPU sharding of the two paths: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

  Path B (pairs)   fully independent units: pair p -> rank p mod G, fragments replicated on every GPU,
                   results gathered on rank 0 in the original pair order.  NO data-path collective.
  Path A (frames)  contiguous frame blocks per rank into private volumes, then ONE exchange step:
                   all-gather of the touched unit keys -> union, ONE reduce(sum) to rank 0 (or all-reduce) over
                   the [key][sdf*weight | weight] planes of the union, per-voxel divide on import.
                   (The running mean with unit weights is a sum: w = sum_g w_g, sdf = sum_g sdf_g*w_g / w;
                   TSDFVolume.cpp:93-94 applied sequentially gives the same value up to float rounding
                   order, hence tolerance 1e-5 instead of bit parity for this mode.)

The functions only need an object with unit_keys() / export_weighted(keys, ptr) / import_weighted(keys, ptr)
/ synchronize(), so the CPU tests can drive the identical protocol over gloo with a host-memory volume.
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


def union_keys(local_keys, dist, device, pad_to=None):
    """Union of the per-rank touched unit keys (sorted int32 numpy) in ONE fixed-size all-gather (padded with -1): tensor shapes
    are identical on every rank whatever each rank touched.  pad_to = a length every rank knows without talking (the volume's
    max_units: a rank cannot hold more keys than that); without it the ranks first AGREE on the padded length with an
    all_reduce(MAX) of the local counts -- one more collective."""
    import torch
    world = dist.get_world_size()
    keys = np.ascontiguousarray(local_keys, np.int32)
    if pad_to is not None and keys.size <= int(pad_to):
        max_keys = max(int(pad_to), 1)
    else:
        cnt = torch.tensor([keys.size], dtype=torch.int64, device=device)
        dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
        max_keys = max(int(cnt.item()), 1)
    pad = torch.full((max_keys,), -1, dtype=torch.int32)
    if keys.size:
        pad[:keys.size] = torch.from_numpy(keys)
    pad = pad.to(device)
    allk = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(allk, pad)                               # (list form: also supported by gloo in the CPU tests)
    u = torch.unique(torch.stack(allk))
    return u[u >= 0].to(torch.int32).cpu().numpy()


def merge_volumes(vol, dist, device, sync_stream=None, mode="reduce", root=0):
    """Frame-split merge.  mode "reduce" (default): ONE reduce(sum) to `root` -- the "final reduce of per-GPU TSDF
    volume-unit weights" of BASELINE.json; afterwards `root` holds the complete volume (the other ranks keep
    their partial volumes).  mode "all_reduce": every rank ends with the complete volume (about 1.75x the
    link traffic of the reduce on a ring).  Returns the union size.
    sync_stream: callable that orders the communication stream after the volume's kernels and vice versa.  None (default)
    is SAFE for any stream set-up: the volume's streams are drained after the export and torch's current stream (the one
    the collective is ordered on) is drained before the import -- two host waits, once per job."""
    import torch
    union = union_keys(vol.unit_keys(), dist, device, pad_to=getattr(vol, "max_units", None))
    if union.size == 0:
        return 0
    buf = torch.empty((union.size, 2, 64 ** 3), dtype=torch.float32, device=device)
    vol.export_weighted(union, buf.data_ptr())
    if sync_stream:
        sync_stream()
    else:
        vol.synchronize()                                    # k_export_weighted has written buf
    if mode == "all_reduce":
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)           # the ONLY data-path collective of the pipeline
    else:
        dist.reduce(buf, dst=root, op=dist.ReduceOp.SUM)
    if sync_stream:
        sync_stream()
    elif getattr(device, "type", str(device)) != "cpu" and str(device) != "cpu":
        torch.cuda.current_stream(device).synchronize()      # the reduce has landed in buf
    if mode == "all_reduce" or dist.get_rank() == root:
        vol.import_weighted(union, buf.data_ptr())
    vol.synchronize()
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
        """root < 0: merged volume on every rank; else only on `root`.  Returns the size of the key union."""
        import ctypes as C
        from . import _ffi
        n = C.c_int(0)
        _ffi.check(self._lib.er_tsdf_allreduce(vol._h, self._h, int(root), C.byref(n)), "er_tsdf_allreduce")
        return n.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.er_comm_destroy(self._h)
            self._h = None
