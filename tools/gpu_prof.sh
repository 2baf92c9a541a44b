#!/bin/bash
# rocprofv3 kernel trace of the bench (run via gpurun).  Args: rows steps tag
set -o pipefail
ROWS=${1:-1000000}; STEPS=${2:-50}; TAG=${3:-prof}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/$TAG -o $TAG -- python3 $REPO/bench.py --rows $ROWS --steps $STEPS --warmup 5 --no-cpu-baseline > $REPO/gpurun_out/$TAG/bench.log 2>&1
rc=$?
tail -2 $REPO/gpurun_out/$TAG/bench.log
find $REPO/gpurun_out/$TAG -name "*stats*" | head
exit $rc
