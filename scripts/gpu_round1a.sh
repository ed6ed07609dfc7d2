#!/bin/bash
# First GPU contact: TSDF parity tests, short bench, rocprof kernel stats.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "gfx|Compute Unit" | head -4 > gpurun_out/rocminfo.txt 2>&1
nproc > gpurun_out/nproc.txt; lscpu | grep "Model name" >> gpurun_out/nproc.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py --steps 10 --warmup 1 --cpu-sample 50 > gpurun_out/bench_short.log 2>&1; tail -2 gpurun_out/bench_short.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof1 -o r01a -- python $OLDPWD/bench.py --steps 6 --warmup 1 --cpu-sample 0 > $OLDPWD/gpurun_out/rocprof_run.log 2>&1
cd $OLDPWD; find /tmp/prof1 -name "*stats*" | head; for f in $(find /tmp/prof1 -name "*kernel_stats*.csv"); do cp $f gpurun_out/; done
ls gpurun_out
