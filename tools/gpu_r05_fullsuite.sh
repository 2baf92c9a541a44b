#!/bin/bash
# the whole GPU suite (what the driver runs at round end), log under gpurun_out/
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r05_gpu_suite.log
timeout -k 10 1100 python -m pytest tests/ -m gpu -q -p no:cacheprovider --durations=15 > $L 2>&1; rc=$?
tail -30 $L | cut -c1-250
exit $rc
