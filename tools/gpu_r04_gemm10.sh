#!/bin/bash
# k_gemm10_tn (kind 11) against k_gemm9_tn (10), the 8-phase kernel (7) and the vendor library: parity (max_err) + time
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_gemm10.log
: > $L
for epi in 0 1 2; do
  shapes="51200x2304x768,51200x768x768,51200x3072x768,51200x768x3072,12800x768x3072,6656x2304x768"
  echo "== epi $epi" >> $L
  timeout -k 10 300 python tools/bench_gemm.py --kind 10,11,7 --epi $epi --shapes $shapes >> $L 2>&1 || { tail -20 $L; exit 1; }
done
grep -E "^==|^\{" $L | cut -c1-230
