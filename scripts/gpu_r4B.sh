#!/bin/bash
# round 4: why does the bench see 6.3 ms for the batch cloud build where scripts/cloud_build_probe.py sees 3.7 ms?  (resident clouds / no pause between the calls)
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; export TMPDIR=/tmp
for res in 0 1; do for ns in 0 1; do ER_CBP_RESIDENT=$res ER_CBP_NOSLEEP=$ns timeout 120 python scripts/cloud_build_probe.py 25 250000 8 2>&1 | tail -1; done; done
