#!/bin/bash
# round 4: k_gemm9_tn after the in-kernel K-cut of the remainder tiles was measured (profiles/r04_gemm9_tail_slices.log) and removed:
# GEMM + encoder parity, then the 100-pair forward with the persistent kernel (default) and without it
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_gemm9_final.log
: > $L
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider -x >> $L 2>&1; rc=$?
tail -3 $L
[ $rc -ne 0 ] && tail -40 $L && exit $rc
echo "== forward" | tee -a $L
for shape in xlmr-base xlmr-large; do
  echo "default $shape" >> $L
  timeout -k 10 200 python tools/bench_rerank.py --shape $shape >> $L 2>&1 || exit $?
  echo "8p $shape" >> $L
  VF_GEMM_9=0 timeout -k 10 200 python tools/bench_rerank.py --shape $shape >> $L 2>&1 || exit $?
done
grep -E "^\{|^8p|^default|==" $L | cut -c1-260
