#!/bin/bash
# Round 5: the loopback merge on the GPU -- the new four-rank test, bin/Integrate --gpus 3 --same_device, and the 8-rank merge of configs[3] inside the merge plan probe.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; SECONDS=0
timeout 500 python -m pytest tests/test_tsdf_gpu.py tests/test_host_programs_gpu.py tests/test_fopt_gpu.py -q -m gpu -p no:cacheprovider -s --tb=short -k "loopback or multi_gpu_modes or frame_split or one_rank or hand_off" > gpurun_out/r5i_pytest.log 2>&1; echo "pytest exit $? t=${SECONDS}s"
grep -a "loopback merge\|passed\|failed\|Error\|assert \|^E " gpurun_out/r5i_pytest.log | cut -c1-600 | tail -12
timeout 400 python scripts/merge_plan_probe.py > gpurun_out/r5i_merge_plan.txt 2>&1; echo "merge plan exit $? t=${SECONDS}s"; grep -a "loopback\|export" gpurun_out/r5i_merge_plan.txt | cut -c1-900
echo "== done t=${SECONDS}s"
