#!/bin/bash
# A/B of k_integrate variants inside ONE gpurun call (same box, interleaved): prints integrate ms/launch per variant.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
for rep in 1 2; do
for V in "$@"; do
  ER_TSDF_VARIANT=$V timeout 300 python bench.py --steps 12 --warmup 2 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('variant %3s  value %8.0f fps  ms/step %.3f  k_integrate %.3f ms  frac %.3f  upd %.0f' % ('$V', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], r['voxel_updates']))"
done; done | tee gpurun_out/ab.txt
