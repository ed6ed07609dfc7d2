for q in 4 8 12 16 24; do
  echo "== GPU_MAX_HW_QUEUES=$q ER_ICP_DIRECT_LISTS=0"; env GPU_MAX_HW_QUEUES=$q ER_ICP_DIRECT_LISTS=0 timeout 300 python scripts/icp_realistic_probe.py 2>/dev/null | head -2
done
