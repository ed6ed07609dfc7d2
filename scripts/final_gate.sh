#!/bin/bash
# THE LAST GPU ACTION OF A ROUND (run HERE, in the build container): the driver's own gate -- `pytest tests -x -q -m gpu` in the driver's
# order, then smoke(), then the default bench -- on the COMMITTED head, in one gpurun call.
#   * refuses a dirty tree: what is validated must be what is committed (round 3 committed a -m gpu test nine minutes AFTER its last
#     full GPU run; the driver's -x stopped at it and 36 parity tests of the hot path never ran);
#   * stamps HEAD into .gate_head (git-ignored, travels with the snapshot) and scripts/gpu_final.sh copies it into every log it writes;
#   * afterwards: `git status` must still be clean and HEAD unchanged, or the validation does not count.
# Rule that goes with it: after this script has run, NOTHING under tests/ or elasticreconstruction_amd/ is committed any more (docs and
# profiles/ only).   usage: bash scripts/final_gate.sh <tag> [seconds for the GPU command, default 900]
set -u
cd "$(dirname "$0")/.."
TAG="${1:?usage: final_gate.sh <tag> [seconds]}"; LIMIT="${2:-900}"
if [ -n "$(git status --porcelain)" ]; then
  echo "final_gate: the tree is dirty -- commit (or stash) first:"; git status --short; exit 1
fi
python -c "import __graft_entry__ as g; g.build()" > /tmp/final_gate_build.log 2>&1 || { echo "final_gate: build() failed"; tail -20 /tmp/final_gate_build.log; exit 1; }
if [ -n "$(git status --porcelain)" ]; then echo "final_gate: build() changed tracked files"; git status --short; exit 1; fi
git rev-parse HEAD > .gate_head
echo "final_gate: HEAD $(cat .gate_head)  tag $TAG"
/usr/local/graft/bin/gpurun --timeout "$LIMIT" -- "GATE_ONLY=${GATE_ONLY:-0} GATE_TESTS=\"${GATE_TESTS:-tests}\" bash scripts/gpu_final.sh $TAG $LIMIT"
rc=$?
echo "final_gate: gpurun exit $rc; logs: gpurun_out/pytest_gpu_$TAG.log gpurun_out/smoke_$TAG.log gpurun_out/bench_default_$TAG.json"
[ "$(git rev-parse HEAD)" = "$(cat .gate_head)" ] || echo "final_gate: WARNING: HEAD moved during the run"
exit $rc
