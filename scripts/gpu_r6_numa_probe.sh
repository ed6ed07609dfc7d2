# Round 6: does the slow mode of the list copies (20 against 51 GB/s) come from WHERE the page-locked buffer lives?  The kinfu-like list in a fresh process,
# copies (ER_ICP_DIRECT_LISTS=0) and in-place writes, with the process bound to the cores of each NUMA node in turn (first touch places the page-locked arena).
echo "nodes: $(ls -d /sys/devices/system/node/node* 2>/dev/null | wc -l); GPU numa_node: $(cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' ')"
rocm-smi --showtoponuma 2>/dev/null | grep -i "numa" | head -4
for n in $(ls -d /sys/devices/system/node/node* | sed 's/.*node//' | head -4); do
  cpus=$(cat /sys/devices/system/node/node$n/cpulist)
  for mode in "ER_ICP_DIRECT_LISTS=0" "ER_ICP_DIRECT_LISTS=1"; do
    echo "== node $n (cpus $cpus) $mode"
    taskset -c $cpus env $mode timeout 300 python scripts/icp_realistic_probe.py 2>/dev/null | head -2
  done
done
echo "== unbound ER_ICP_DIRECT_LISTS=0"; env ER_ICP_DIRECT_LISTS=0 timeout 300 python scripts/icp_realistic_probe.py 2>/dev/null | head -2
