#!/bin/bash
# the whole GPU suite as the driver runs it, timed
set -o pipefail
mkdir -p gpurun_out
t0=$(date +%s)
timeout -k 10 1100 python3 -m pytest tests/ -x -q -m gpu --durations=25 > gpurun_out/r06_suite.log 2>&1
rc=$?
echo "rc $rc wall $(( $(date +%s) - t0 )) s"
tail -45 gpurun_out/r06_suite.log | cut -c1-200
exit $rc
