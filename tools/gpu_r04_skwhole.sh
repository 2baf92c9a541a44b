#!/bin/bash
# products of less than one round of 256 x 256 tiles with a long K (FFN-down of a 13- / 25-pair batch): the 8-phase kernel with every
# tile cut along K (split mode 2) against the dispatch's choice, the persistent kernel, the 128 x 128 kernel and the vendor library
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_skwhole.log
: > $L
S=6656x768x3072,6656x1024x4096,12800x768x3072,12800x1024x4096,3328x768x3072
for mode in 0 2; do
  echo "== VF_SK_MODE=$mode epi 2" >> $L
  VF_SK_MODE=$mode timeout -k 10 200 python tools/bench_gemm.py --kind 0,7,10,3 --epi 2 --iters 50 --shapes $S >> $L 2>&1 || { tail -20 $L; exit 1; }
done
grep -E "^==|^\{" $L | cut -c1-200
