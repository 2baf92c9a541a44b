#!/bin/bash
# round 4: k_gemm9_tn with desynchronised workgroups (start stagger 0 / 50 / 100 / 150 % of a tile's time), nt stores
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_gemm9c.log
: > $L
for stg in 0 50 100 150; do
  echo "== isolated products, kind 10, stagger $stg" | tee -a $L
  VF_GEMM_9_STAGGER=$stg VF_SK_MODE=0 timeout -k 10 300 python tools/bench_gemm.py --kind 10 --epi 0 >> $L 2>&1 || exit $?
done
echo "== isolated products, kind 10 nt stores, stagger 0 / 100" | tee -a $L
for stg in 0 100; do
VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_nt.so VF_GEMM_9_STAGGER=$stg VF_SK_MODE=0 timeout -k 10 300 python tools/bench_gemm.py --kind 10 --epi 0 >> $L 2>&1 || exit $?
done
echo "== kind 7 for reference" | tee -a $L
VF_SK_MODE=0 timeout -k 10 300 python tools/bench_gemm.py --kind 7 --epi 0 >> $L 2>&1 || exit $?
echo "== forward" | tee -a $L
for shape in xlmr-base xlmr-large; do
  echo "8p $shape" >> $L
  VF_GEMM_9=0 timeout -k 10 200 python tools/bench_rerank.py --shape $shape >> $L 2>&1 || exit $?
  for stg in 0 100; do
    echo "gemm9 stagger $stg $shape" >> $L
    VF_GEMM_9=1 VF_GEMM_9_STAGGER=$stg timeout -k 10 200 python tools/bench_rerank.py --shape $shape >> $L 2>&1 || exit $?
  done
  echo "gemm9 nt stagger 100 $shape" >> $L
  VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_nt.so VF_GEMM_9=1 VF_GEMM_9_STAGGER=100 timeout -k 10 200 python tools/bench_rerank.py --shape $shape >> $L 2>&1 || exit $?
done
grep -E "^\{|^8p|^gemm9|==" $L | cut -c1-200
