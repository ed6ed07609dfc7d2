#!/bin/bash
# configs[3] / configs[4] at one GPU's share + the PCIe-inclusive variant of configs[1]: smoke of the paths the default run does not take
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out
for a in "--config 4" "--config 5" "--host-input"; do
  echo "=== bench.py $a"
  timeout 400 python bench.py $a --cpu-sample 0 --min-seconds 0.2 2> gpurun_out/cfg_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(json.dumps({k: d[k] for k in ('value','ms_per_step','n_gpus','scaling')}), d['config']['workload'][:90])
print('  units', d['config'].get('volume_units_touched'), 'merge_union', d['config'].get('merge_union_units'), 'roofline.frac', d['roofline'].get('frac'))
i = d.get('icp')
if i: print('  icp', {k: i[k] for k in i if k in ('pairs_per_s','pairs_total','pairs','accepted','rejected_precheck','parity_checked')})
" || tail -5 gpurun_out/cfg_err.txt
done
