#!/bin/bash
set -o pipefail
ROWS=${1:-1000000}; STEPS=${2:-30}; TAG=${3:-solo}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/$TAG -o $TAG -- python3 $REPO/tools/solo.py $ROWS $STEPS > $REPO/gpurun_out/$TAG/run.log 2>&1
rc=$?
tail -2 $REPO/gpurun_out/$TAG/run.log
exit $rc
