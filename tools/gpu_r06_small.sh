#!/bin/bash
# Round 6, item 1: the 8-GPU rank's scan (1.25M x 768) and configs[1] (1M x 768): CU-split / sample / depth sweep on a side stream
set -o pipefail
mkdir -p gpurun_out
for spec in "1250000 768" "1000000 768" "1250000 1024"; do
  set -- $spec
  timeout -k 10 300 python3 tools/r06_small_sweep.py $1 $2 > gpurun_out/r06_small_sweep_$1x$2.log 2>&1 || { tail -5 gpurun_out/r06_small_sweep_$1x$2.log; exit 1; }
  cat gpurun_out/r06_small_sweep_$1x$2.log
done
