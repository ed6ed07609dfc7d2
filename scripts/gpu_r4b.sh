cd "${GRAFT_REPO_ROOT:-/root/repo}"
GATE_X="" GATE_BENCH=0 GATE_ONLY=1 bash scripts/gpu_final.sh r04b 420 2>&1 | tail -25
echo "=== ICP PMC before"
bash scripts/gpu_icp_pmc.sh r04b_before 50 3
