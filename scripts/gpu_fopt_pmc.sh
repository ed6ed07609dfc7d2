#!/bin/bash
# PMC pass (kernel-trace + counters only) of the FragmentOptimizer Gram kernels: FP64 matrix-core instruction counts and busy cycles.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/pmc_fopt; export TMPDIR=/tmp; cd /tmp
cat > /tmp/fopt_pmc_driver.py <<'PY'
import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from fopt_helpers import make_scene
from elasticreconstruction_amd.fopt import FragmentOptimizer
sc = make_scene(num=4, n=120000)
g = FragmentOptimizer(4, 8, 3.0)
for f, (x, n) in enumerate(sc["frags"]):
    g.SetCloud(f, x, n); g.UpdatePose(f, sc["poses"][f].astype(np.float32))
print("groups", g.SetCorrespondences(sc["pairs"]), "correspondences", sum(p[2].shape[0] for p in sc["pairs"]))
Rt = np.stack([P[:3, :3].T.reshape(9) for P in sc["poses"]])
for _ in range(3):
    g.AssembleSLAC(Rt); g.AssembleRigid(); g.AssembleNonrigid(1.0)
PY
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /tmp/pmc_fopt -o p -- python /tmp/fopt_pmc_driver.py > $R/gpurun_out/pmc_fopt/run.log 2>&1
for f in $(find /tmp/pmc_fopt -name "*counter_collection.csv"); do cp "$f" $R/gpurun_out/pmc_fopt/pass1_counter_collection.csv; done
cd $R; python scripts/pmc_summary.py gpurun_out/pmc_fopt > gpurun_out/pmc_fopt/summary.txt 2>&1; grep -A8 "k_fopt_gram" gpurun_out/pmc_fopt/summary.txt | head -40; tail -2 gpurun_out/pmc_fopt/run.log
