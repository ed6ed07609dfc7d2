#!/bin/bash
# One-off instrumentation run: needs liber_hip.so built with -DER_STATS (see scripts/build_stats.sh).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
python - <<'PY'
import ctypes as C, numpy as np, torch, sys
sys.path.insert(0, '.')
from elasticreconstruction_amd import synth, _ffi
from elasticreconstruction_amd.tsdf import TSDFVolume
L = _ffi.lib()
sc = synth.make_scenario(600, interval=50, warp=True, device="cuda:0", total_frames=3000, revolutions=1.0)
w = synth.warp_arrays(sc)
vol = TSDFVolume(max_units=640)
out = (C.c_ulonglong * 4)()
L.er_debug_stats(out, 1)
for s in range(12):
    lo, hi = s*50, s*50+50
    ws = dict(ctr=w["ctr"][s:s+1], resolution=8, length=np.float32(3.0), grid_index=np.zeros(50, np.int32), seg=w["seg"][lo:hi], madj=w["madj"][lo:hi])
    vol.IntegrateFrames(None, sc["traj"][lo:hi], ws, device_ptr=sc["depth"].data_ptr() + lo*307200*2)
vol.synchronize()
L.er_debug_stats(out, 0)
rf, rfu, upd = out[0], out[1], out[2]
print("row-frames %d, with >=1 update %d (%.1f%%), voxel updates %d (%.1f%% of lanes in visited rows; %.1f%% of lanes in updating rows)" % (rf, rfu, 100.0*rfu/rf, upd, 100.0*upd/(rf*64), 100.0*upd/(rfu*64)))
PY
