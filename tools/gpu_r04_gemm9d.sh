#!/bin/bash
# round 4: k_gemm9_tn with the epilogue split into two halves inside the MFMA phases
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_gemm9d.log
: > $L
echo "== parity (kind 10)" | tee -a $L
timeout -k 10 300 python -m pytest tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider -x -k "gemm_kernels_match_torch and 10" >> $L 2>&1; rc=$?
tail -3 $L
[ $rc -ne 0 ] && tail -40 $L && exit $rc
echo "== stamps" | tee -a $L
for stg in 0 100; do
  VF_GEMM_9_STAGGER=$stg timeout -k 10 200 python tools/gemm9_stamps.py --shapes 51200x2304x768 >> $L 2>&1 || { tail -20 $L; exit 1; }
done
VF_GEMM_9_STAGGER=0 timeout -k 10 200 python tools/gemm9_stamps.py --shapes 51200x768x768 --epi 2 >> $L 2>&1
VF_GEMM_9_STAGGER=0 timeout -k 10 200 python tools/gemm9_stamps.py --shapes 51200x3072x768 --epi 1 >> $L 2>&1
for stg in 0 100; do
for epi in 0 1 2; do
  echo "== isolated, epi $epi, kinds 7 / 10 (stagger $stg)" | tee -a $L
  VF_GEMM_9_STAGGER=$stg timeout -k 10 300 python tools/bench_gemm.py --kind 7,10 --epi $epi >> $L 2>&1 || exit $?
done
done
echo "== forward" | tee -a $L
for shape in xlmr-base xlmr-large; do
  echo "8p $shape" >> $L
  VF_GEMM_9=0 timeout -k 10 200 python tools/bench_rerank.py --shape $shape >> $L 2>&1 || exit $?
  for stg in 0 100; do
    echo "gemm9 stagger $stg $shape" >> $L
    VF_GEMM_9=1 VF_GEMM_9_STAGGER=$stg timeout -k 10 200 python tools/bench_rerank.py --shape $shape >> $L 2>&1 || exit $?
  done
done
grep -E "^\{|^8p|^gemm9|==" $L | cut -c1-1200
