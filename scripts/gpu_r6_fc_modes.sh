# Round 6: FindCorrespondence with lists written in place: the group cut into ER_ICP_FC_SPLIT parts on two compute streams (scripts/icp_realistic_probe.py, first 3 lines)
for mode in "ER_ICP_FC_SPLIT=1" "ER_ICP_FC_SPLIT=2" "ER_ICP_FC_SPLIT=4" "ER_ICP_FC_SPLIT=8" "ER_ICP_FC_SPLIT=1"; do
  echo "== $mode"; env $mode timeout 300 python scripts/icp_realistic_probe.py 2>/dev/null | head -3
done
