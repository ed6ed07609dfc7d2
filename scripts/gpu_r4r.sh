#!/bin/bash
# round 4, call 8: the evidence files of the final tree -- rocprofv3 --kernel-trace --stats of a bench run, the PMC passes (separate runs)
# of the same command -> gpurun_out/prof_r04r/, gpurun_out/pmc_r04r/summary.txt; plus the ICP kernel stats.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
ARGS="--steps 20 --warmup 1 --no-alone --no-streamed --cpu-sample 0 --icp-pairs 0 --other-configs 0 --min-seconds 0.05 --max-passes 1"
bash scripts/gpu_prof.sh r04r $ARGS > gpurun_out/prof_r04r.log 2>&1; python scripts/kstats.py gpurun_out/prof_r04r/r04r_kernel_stats.csv 2>&1 | head -12; cat gpurun_out/prof_r04r/bench.json | cut -c1-300
bash scripts/gpu_pmc.sh r04r $ARGS > gpurun_out/pmc_r04r.log 2>&1; grep -A14 "^k_integrate\|^k_prepare\|^k_reproject_scatter" gpurun_out/pmc_r04r/summary.txt | head -60
rm -f gpurun_out/pmc_r04r/pass*_counter_collection.csv
export TMPDIR=/tmp; cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_icp4 -o icp -- python $GRAFT_REPO_ROOT/scripts/icp_list_probe.py 50 6 > $GRAFT_REPO_ROOT/gpurun_out/prof_icp_r04r.log 2>&1
cd $GRAFT_REPO_ROOT; for f in $(find /tmp/prof_icp4 -name "*kernel_stats.csv"); do cp "$f" gpurun_out/r04r_icp_kernel_stats.csv; done
python scripts/kstats.py gpurun_out/r04r_icp_kernel_stats.csv | grep -E "k_count|k_icp|k_find|k_scan|k_compact|k_grid|DeviceRadix|rocprim" | head -20
tail -6 gpurun_out/prof_icp_r04r.log
