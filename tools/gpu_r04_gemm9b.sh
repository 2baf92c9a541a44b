#!/bin/bash
# round 4: k_gemm9_tn A/B -- MFMA operand order (default build: W fragment shared; libvf_o0.so: A fragment shared), and the
# per-tile fixed cost of kinds 7 / 10 from a sweep over K at 600 tiles (split-K tail off)
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_gemm9b.log
: > $L
echo "== K sweep, 51200 x 768 x K, bias epilogue, split-K tail off" | tee -a $L
VF_SK_MODE=0 timeout -k 10 300 python tools/bench_gemm.py --kind 7,10 --epi 0 --shapes 51200x768x256,51200x768x512,51200x768x768,51200x768x1536,51200x768x3072 >> $L 2>&1 || exit $?
echo "== forward, default kernels / gemm9 order 1 (default lib) / gemm9 order 0 (libvf_o0)" | tee -a $L
for shape in xlmr-base xlmr-large; do
  for rep in 1 2; do
    echo "8p $shape" >> $L
    VF_GEMM_9=0 timeout -k 10 200 python tools/bench_rerank.py --shape $shape >> $L 2>&1 || exit $?
    echo "gemm9-order1 $shape" >> $L
    VF_GEMM_9=1 timeout -k 10 200 python tools/bench_rerank.py --shape $shape >> $L 2>&1 || exit $?
    echo "gemm9-order0 $shape" >> $L
    VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_o0.so VF_GEMM_9=1 timeout -k 10 200 python tools/bench_rerank.py --shape $shape >> $L 2>&1 || exit $?
  done
done
echo "== chain 1024/4096" | tee -a $L
timeout -k 10 200 python tools/bench_gemm_chain.py --kind 10 --hidden 1024 --ffn 4096 --rows 51200 >> $L 2>&1
timeout -k 10 200 python tools/bench_gemm_chain.py --kind 7 --hidden 1024 --ffn 4096 --rows 51200 >> $L 2>&1
grep -E "^\{|^8p|^gemm9|==" $L | cut -c1-330
