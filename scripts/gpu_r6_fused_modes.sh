for mode in "ER_ICP_FC_SPLIT=4" "ER_ICP_FC_SPLIT=1" "ER_ICP_DIRECT_LISTS=0" "ER_ICP_DIRECT_LISTS=d"; do
  echo "== $mode"; env $mode ER_PROBE_HARD=0 ER_PROBE_CLOUDS=0 ER_PROBE_SHARES="1 3 6" timeout 300 python scripts/icp_list_probe.py 50 12 2>/dev/null | grep -v "^$" | cut -c1-230
done
