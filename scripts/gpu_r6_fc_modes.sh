# Round 6: how the correspondence lists of a 50-pair list reach page-locked host memory (scripts/icp_realistic_probe.py, fresh process each; first 3 lines)
for mode in "ER_ICP_DIRECT_LISTS=1" "ER_ICP_DIRECT_LISTS=2" "ER_ICP_DIRECT_LISTS=1 ER_HIP_LIB=$PWD/elasticreconstruction_amd/_ab/liber_hip_narrowcompact.so" "ER_ICP_DIRECT_LISTS=0 ER_ICP_COPY_STREAMS=2" "ER_ICP_DIRECT_LISTS=2"; do
  echo "== $mode"; env $mode timeout 300 python scripts/icp_realistic_probe.py 2>/dev/null | head -3
done
