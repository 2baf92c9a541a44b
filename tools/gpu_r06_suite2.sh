#!/bin/bash
# the rest of the GPU suite from the failing test on + the attention PMC passes
set -o pipefail
mkdir -p gpurun_out
t0=$(date +%s)
timeout -k 10 1100 python3 -m pytest tests/ -x -q -m gpu --durations=12 > gpurun_out/r06_suite.log 2>&1
rc=$?
echo "rc $rc wall $(( $(date +%s) - t0 )) s"
tail -30 gpurun_out/r06_suite.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
bash tools/gpu_r06_att_pmc.sh
