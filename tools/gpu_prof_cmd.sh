#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python tool: gpu_prof_cmd.sh <tag> <script> [args...]
set -o pipefail
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/$TAG
SCRIPT=$REPO/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/$TAG -o $TAG -- python3 $SCRIPT "$@" > $REPO/gpurun_out/$TAG/run.log 2>&1
rc=$?
tail -2 $REPO/gpurun_out/$TAG/run.log
exit $rc
