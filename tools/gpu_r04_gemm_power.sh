#!/bin/bash
# the same products on random operands, on all-zero operands and on constant operands: how much of the gap to the MFMA peak is the
# clock giving way under the power of toggling operand bits
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_gemm_power.log
: > $L
for data in random zeros ones random; do
  echo "== data $data" >> $L
  timeout -k 10 200 python tools/bench_gemm.py --kind 10,11,7 --epi 0 --data $data --iters 50 --shapes 51200x2304x768,51200x768x3072 >> $L 2>&1 || { tail -20 $L; exit 1; }
  VF_G10_ABL=15 timeout -k 10 120 python tools/bench_gemm.py --kind 11 --epi 0 --check 0 --data $data --iters 50 --shapes 51200x768x3072 >> $L 2>&1 || { tail -20 $L; exit 1; }
done
grep -E "^==|^\{" $L | cut -c1-170
