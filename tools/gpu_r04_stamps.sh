#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_gemm9_stamps.log
: > $L
for stg in 0 100; do
  VF_GEMM_9_STAGGER=$stg timeout -k 10 200 python tools/gemm9_stamps.py >> $L 2>&1 || { tail -20 $L; exit 1; }
done
VF_GEMM_9_STAGGER=0 timeout -k 10 200 python tools/gemm9_stamps.py --epi 2 --shapes 51200x768x768 >> $L 2>&1
cat $L
