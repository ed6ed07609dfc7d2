"""Stress of bin/FragmentOptimizer's non-rigid mode for the intermittent 'not positive definite' failure seen once under
pytest -n 4 (gpurun call r02l): the test's own dataset, N runs of the program while other processes keep the GPU busy."""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fopt_helpers import make_scene
from test_fopt_oracle import _write_dataset
BIN = os.path.join(ROOT, "elasticreconstruction_amd", "bin", "FragmentOptimizer")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
hogs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
d = tempfile.mkdtemp()
_write_dataset(make_scene(num=3, n=3000, res=4), d)
args = ["--num", "3", "--resolution", "4", "--length", "3.0", "--weight", "1.7", "--inner_iteration", "2", "--iteration", "1"]
bg = [subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "1", "--cpu-sample", "0", "--icp-pairs", "0",
                        "--no-streamed", "--min-seconds", "60"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for _ in range(hogs)]
time.sleep(25)
PAR = int(sys.argv[3]) if len(sys.argv) > 3 else 1           # copies of the program running at the same time (like pytest -n 4)
from concurrent.futures import ThreadPoolExecutor
t0 = time.time()


def one(i):
    out = os.path.join(d, "o%d.ctr" % (i % PAR))
    cmd = [BIN] + args + ["--registration", os.path.join(d, "reg_output.log"), "--dir", d + "/", "--rgbdslam", os.path.join(d, "rgbd.log"),
                          "--interval", "1", "--blacklistpair", "0", "--save_to", out]
    r = subprocess.run(cmd, cwd=d, capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        print("run %d FAILED: %s" % (i, r.stderr.strip()[-200:]), flush=True)
        return None
    if "not positive" in r.stderr or "again" in r.stderr:
        print("run %d NOTE on stderr: %s" % (i, r.stderr.strip()[-200:]), flush=True)
    return np.loadtxt(out)


ref, bad, worst = None, 0, 0.0
with ThreadPoolExecutor(PAR) as ex:
    for lo in range(0, N, PAR):
        for c in ex.map(one, range(lo, min(lo + PAR, N))):
            if c is None:
                bad += 1
                continue
            if ref is None:
                ref = c
            worst = max(worst, float(np.abs(c - ref).max()))
for p in bg:
    p.kill()
print("runs %d (%d at a time)  failed %d  max |ctr - first| %.3e  (%.1f s, %d background benches)" % (N, PAR, bad, worst, time.time() - t0, hogs))
