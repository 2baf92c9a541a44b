#!/bin/bash
# PMC pass (own run, kernel-trace only alongside): HBM traffic of the scan kernel.  Args: rows steps tag counters
set -o pipefail
ROWS=${1:-10000000}; STEPS=${2:-6}; TAG=${3:-pmc}; CTR=${4:-FETCH_SIZE}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d $REPO/gpurun_out/$TAG -o $TAG -- python3 $REPO/bench.py --rows $ROWS --steps $STEPS --warmup 2 --no-cpu-baseline --no-rerank > $REPO/gpurun_out/$TAG/bench.log 2>&1
rc=$?
tail -2 $REPO/gpurun_out/$TAG/bench.log | cut -c1-300
ls $REPO/gpurun_out/$TAG
exit $rc
