"""bench_dry.py -- the stand-ins `bench.py --dry-run` uses to rehearse its control flow on a machine without GPUs (gloo instead of RCCL,
a host-array volume).  Nothing here measures anything; tests/test_distributed_cpu.py launches it with the driver's command line."""
import ctypes

class DryVolume:
    """--dry-run stand-in for elasticreconstruction_amd.tsdf.TSDFVolume: a host-array volume with the handful of methods main()
    calls and the unit_keys / export_weighted / import_weighted / synchronize surface parallel.merge_volumes drives (the same
    surface tests/test_distributed_cpu.py's HostVolume has).  Every rank touches one unit of its own per 64-frame launch plus
    one unit shared by all ranks, with unit weights, so the merge has a real union, overlapping and private keys, and exact sums
    to check.  NO arithmetic of the hot path happens here: a dry run measures nothing, it rehearses bench.py's control flow --
    argument handling, rank gating, the collective sequence, the JSON line -- on a machine without GPUs."""
    VOX = 64 ** 3

    def __init__(self, rank, max_units):
        import numpy as np
        self._np, self.rank, self.max_units = np, rank, max_units
        self.units, self._frames, self._launches = {}, 0, 0

    def set_stream(self, _):
        pass

    def synchronize(self):
        pass

    def set_profiling(self, _):
        self._launches = 0

    def get_profile(self):
        return {"launches": self._launches, "integrate_ms": 0.25 * self._launches, "unit_visits": 16 * self._launches}

    def reset(self):
        self.units, self._frames = {}, 0

    def _add(self, key, n):
        np = self._np
        if key not in self.units:
            self.units[key] = (np.ones(self.VOX, np.float32), np.zeros(self.VOX, np.float32))
        self.units[key][1][:] += np.float32(n)

    def IntegrateFrames(self, depth, T, warp=None, device_ptr=None):
        n = len(T)
        for lo in range(0, n, 64):
            m = min(64, n - lo)
            self._add(131329 + 1000 * (self.rank + 1) + (self._frames // 64) % 8, m)       # a private unit of this rank
            self._add(131329, m)                                                            # ... and one every rank touches
            self._frames += m
            self._launches += 1

    def unit_count(self):
        return len(self.units)

    def unit_keys(self):
        return self._np.array(sorted(self.units), self._np.int32)

    def sum_weight(self):
        return float(sum(float(w[0]) * self.VOX for _, w in self.units.values()))

    def _view(self, ptr, n):
        return self._np.ctypeslib.as_array((ctypes.c_float * (n * 2 * self.VOX)).from_address(ptr)).reshape(n, 2, self.VOX)

    def export_weighted(self, keys, ptr):
        buf = self._view(ptr, len(keys))
        for q, k in enumerate(keys):
            if int(k) in self.units:
                sdf, w = self.units[int(k)]
                buf[q, 0], buf[q, 1] = sdf * w, w
            else:
                buf[q] = 0

    def import_weighted(self, keys, ptr):
        np = self._np
        buf = self._view(ptr, len(keys))
        for q, k in enumerate(keys):
            sw, w = buf[q, 0].copy(), buf[q, 1].copy()
            with np.errstate(divide="ignore", invalid="ignore"):
                self.units[int(k)] = (np.where(w > 0, sw / w, np.float32(0)).astype(np.float32), w)

    def export_raw(self, keys, ptr):
        buf = self._view(ptr, len(keys))
        for q, k in enumerate(keys):
            buf[q, 0], buf[q, 1] = self.units[int(k)]

    def import_raw(self, keys, ptr):
        buf = self._view(ptr, len(keys))
        for q, k in enumerate(keys):
            self.units[int(k)] = (buf[q, 0].copy(), buf[q, 1].copy())

    def close(self):
        self.units = {}


class DryComm:
    """--dry-run stand-in for parallel.AbiComm: the SAME out-of-band exchange (rank 0's 128-byte id travels by
    broadcast_object_list), then the merge protocol over torch.distributed (gloo) instead of liber_hip.so's RCCL calls."""

    def __init__(self, dist, device):
        self._dist, self._device = dist, device
        box = [bytes(128) if dist.get_rank() else bytes(range(128))]
        dist.broadcast_object_list(box, src=0)
        assert box[0] == bytes(range(128))

    def allreduce(self, vol, root=0):
        from elasticreconstruction_amd import parallel
        return parallel.merge_volumes(vol, self._dist, self._device, root=root)

    def close(self):
        pass
