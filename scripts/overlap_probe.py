import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from elasticreconstruction_amd import synth
from elasticreconstruction_amd.tsdf import TSDFVolume
K, I = 20, 50
sc = synth.make_scenario(K*I, interval=I, warp=True, device="cuda:0", total_frames=3000, revolutions=1.0)
w = synth.warp_arrays(sc); depth = sc["depth"]; px = depth.shape[1]
def ws(s):
    lo, hi = s*I, s*I+I
    return dict(ctr=w["ctr"][s:s+1], resolution=8, length=np.float32(3.0), grid_index=np.zeros(I, np.int32), seg=w["seg"][lo:hi], madj=w["madj"][lo:hi])
def run(vols, reps=2):
    for v in vols:
        for s in range(2): v.IntegrateFrames(None, sc["traj"][s*I:(s+1)*I], ws(s), device_ptr=depth.data_ptr()+s*I*px*2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(K):
        for v in vols:
            v.IntegrateFrames(None, sc["traj"][s*I:(s+1)*I], ws(s), device_ptr=depth.data_ptr()+s*I*px*2)
    for v in vols: v.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return len(vols)*K*I/dt
for n in (1, 2, 1, 2, 3):
    vols = [TSDFVolume(max_units=640) for _ in range(n)]
    print("concurrent volumes %d: aggregate %.0f frames/s" % (n, run(vols)))
    for v in vols: v.close()
