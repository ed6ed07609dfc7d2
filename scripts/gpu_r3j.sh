#!/bin/bash
# Round 3, GPU call J: 8 and 16 register rows per lane in k_integrate -- launch-parameter variants, parity of rows16, VALU counts.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; TAG=r03j; mkdir -p gpurun_out; export TMPDIR=/tmp
SECONDS=0
ER_HIP_LIB=$R/elasticreconstruction_amd/_ab/liber_hip_rows16a.so timeout 600 python -m pytest tests/test_tsdf_gpu.py -m gpu -x -q --tb=short -p no:cacheprovider -k "golden or config2 or randomised or batch_boundary" 2>&1 | tail -3
echo "== t=${SECONDS}s A/B"
bash scripts/ab_libs.sh 2 main rows8b rows8d rows8e rows16a rows16b > gpurun_out/ab_$TAG.txt 2>&1; cat gpurun_out/ab_$TAG.txt
echo "== t=${SECONDS}s pmc"
for v in rows8b rows16a; do
  export ER_HIP_LIB=$R/elasticreconstruction_amd/_ab/liber_hip_$v.so
  OUT=$R/gpurun_out/pmc_${TAG}_$v; mkdir -p $OUT; cd /tmp
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_${TAG}_${v}_1 -o p1 -- python $R/bench.py --steps 20 --warmup 1 --cpu-sample 0 --icp-pairs 0 --no-streamed --no-alone --min-seconds 0.01 > $OUT/run_1.log 2>&1
  for f in $(find /tmp/pmc_${TAG}_${v}_1 -name "*counter_collection.csv"); do cp "$f" $OUT/pass1_counter_collection.csv; done
  cd $R; python scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; rm -f $OUT/pass*_counter_collection.csv
  echo "=== $v"; grep -A9 "^k_integrate" $OUT/summary.txt
done
unset ER_HIP_LIB
echo "== done t=${SECONDS}s"
