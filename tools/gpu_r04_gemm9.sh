#!/bin/bash
# round 4: k_gemm9_tn (persistent, register-direct epilogue): parity, isolated products, dependent chain, the forward
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_gemm9.log
: > $L
echo "== parity (kind 10)" | tee -a $L
timeout -k 10 300 python -m pytest tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider -x -k "gemm_kernels_match_torch and 10" >> $L 2>&1; rc=$?
tail -3 $L
[ $rc -ne 0 ] && tail -40 $L && exit $rc
echo "== isolated products, bias / gelu / residual: kinds 7 (8-phase) and 10 vs vendor" | tee -a $L
for epi in 0 1 2; do
  timeout -k 10 200 python tools/bench_gemm.py --kind 0,7,10 --epi $epi >> $L 2>&1 || exit $?
done
echo "== dependent chain, bias only and forward epilogues" | tee -a $L
for kind in 0 10; do
  timeout -k 10 200 python tools/bench_gemm_chain.py --kind $kind >> $L 2>&1 || exit $?
  timeout -k 10 200 python tools/bench_gemm_chain.py --kind $kind --forward-epilogues >> $L 2>&1 || exit $?
  timeout -k 10 200 python tools/bench_gemm_chain.py --kind $kind --hidden 1024 --ffn 4096 --rows 51200 >> $L 2>&1 || exit $?
done
echo "== forward: re-rank 100 x 512, default vs VF_GEMM_9=1" | tee -a $L
for shape in xlmr-base xlmr-large; do
  for v in 0 1 0 1; do
    echo "VF_GEMM_9=$v $shape" >> $L
    VF_GEMM_9=$v timeout -k 10 200 python tools/bench_rerank.py --shape $shape >> $L 2>&1 || exit $?
  done
done
grep -E "^\{|VF_GEMM_9|==" $L | tail -60
