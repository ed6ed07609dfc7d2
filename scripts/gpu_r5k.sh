#!/bin/bash
# Round 5, after the last gate: (1) the torch.distributed cross-check of the sparse merge (parallel.merge_volumes) with a DEVICE volume, one rank;
# (2) kernel-trace stats and one SQ counter pass of configs[3] (bench.py --config 4: 10 000 frames, 1103 units) -- where that job's time goes.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out; SECONDS=0
COMMON="--cpu-sample 0 --icp-pairs 0 --other-configs 0 --no-streamed --no-alone"
timeout 150 python bench.py --force-merge --merge-impl torch --steps 20 --warmup 1 --min-seconds 0.3 $COMMON > gpurun_out/r05k_force_merge_torch.json 2> gpurun_out/r05k_force_merge_torch.err; echo "force-merge torch exit $? t=${SECONDS}s"
timeout 150 python bench.py --force-merge --merge-impl abi --steps 20 --warmup 1 --min-seconds 0.3 $COMMON > gpurun_out/r05k_force_merge_abi.json 2> gpurun_out/r05k_force_merge_abi.err; echo "force-merge abi exit $? t=${SECONDS}s"
bash scripts/gpu_prof.sh r05k --config 4 --steps 50 --warmup 1 --min-seconds 0.1 $COMMON > /dev/null 2>&1; python scripts/kstats.py gpurun_out/prof_r05k/r05k_kernel_stats.csv 2>&1 | head -14; echo "stats t=${SECONDS}s"
OUT=$R/gpurun_out/pmc_r05k; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout 240 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_r05k_1 -o p1 -- python $R/bench.py --config 4 --steps 50 --warmup 1 --min-seconds 0.01 $COMMON > $OUT/run_1.log 2>&1 )
for f in $(find /tmp/pmc_r05k_1 -name "*counter_collection.csv"); do cp "$f" $OUT/pass1_counter_collection.csv; done
python scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; rm -f $OUT/pass*_counter_collection.csv
grep -A10 "^k_integrate\|^k_reproject_scatter\|^k_prepare\|^k_plan" $OUT/summary.txt | head -60
echo "== done t=${SECONDS}s"
