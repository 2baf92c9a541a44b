#!/bin/bash
# rocprofv3 kernel stats of any python tool.  Usage: gpu_prof_py.sh <tag> <script.py> [args...]
set -o pipefail
TAG=$1; SCRIPT=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/$TAG -o $TAG -- python3 $REPO/$SCRIPT "$@" > $REPO/gpurun_out/$TAG/run.log 2>&1 || { tail -3 $REPO/gpurun_out/$TAG/run.log; exit 1; }
cd $REPO
f=$(find gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/${TAG}_kernel_stats.csv
head -${VF_PROF_LINES:-24} $f | cut -c1-200
find gpurun_out/$TAG -name "*.csv" ! -name "*stats*" -delete
