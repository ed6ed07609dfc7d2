"""End-to-end rate of the drop-in bin/Integrate on configs[1] (3000 frames read from a raw file, warp on, world.pcd written):
what a pipeline script sees, file I/O and PCIe included.  usage: python scripts/gpu_integrate_e2e.py [n_frames]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from elasticreconstruction_amd import synth
from test_host_programs_gpu import write_integrate_inputs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sc = synth.make_scenario(n, interval=50, warp=True, device="cuda:0")
depth = synth.to_numpy_u16(sc["depth"])
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
write_integrate_inputs(d, sc, depth)
args = ["--pose_traj", "pose.log", "--seg_traj", "seg.log", "--ctr", "grids.ctr", "--num", str(n // 50), "--resolution", "8", "--length", "3.0",
        "--interval", "50", "-oni", "frames.raw", "--save_to", "world.pcd", "--max_units", "1024"]
for rep in range(2):
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(ROOT, "elasticreconstruction_amd", "bin", "Integrate")] + args, cwd=d, capture_output=True, text=True)
    dt = time.perf_counter() - t0
    tail = [l for l in r.stdout.splitlines() if "frames" in l.lower() or "written" in l.lower() or "fps" in l.lower()][-4:]
    print("run %d: rc %d, wall %.2f s -> %.0f frames/s end to end (process start, file read, integrate, SaveWorld)" % (rep, r.returncode, dt, n / dt))
    for l in tail: print("   ", l)
