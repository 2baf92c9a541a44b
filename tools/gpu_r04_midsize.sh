#!/bin/bash
# which kernel serves a mid-size product best?  the per-rank batches of a data-parallel re-rank (13 / 25 / 50 pairs x 512 tokens), every kernel forced
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_midsize.log
: > $L
for M in 6656 12800 25600; do
  for epi in 0 2; do
    shapes="${M}x2304x768,${M}x768x768,${M}x3072x768,${M}x768x3072"
    [ $epi = 2 ] && shapes="${M}x768x768,${M}x768x3072"
    for sk in 1 2; do
      echo "== M $M epi $epi VF_SK_MODE=$sk" >> $L
      VF_SK_MODE=$sk timeout -k 10 300 python tools/bench_gemm.py --kind 0,3,5,7,10 --epi $epi --shapes $shapes >> $L 2>&1 || exit 1
    done
  done
done
grep -E "^==|^\{" $L | cut -c1-420
