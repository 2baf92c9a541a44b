#!/bin/bash
# round 4: k_gemm9_tn, epilogue halves placed by wave half (default) vs both in front of the barrier (libvf_nosplit)
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_gemm9e.log
: > $L
echo "== parity (kind 10)" | tee -a $L
timeout -k 10 300 python -m pytest tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider -x -k "gemm_kernels_match_torch and 10" >> $L 2>&1; rc=$?
tail -3 $L
[ $rc -ne 0 ] && tail -40 $L && exit $rc
for lib in default; do
  [ $lib = nosplit ] && export VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_nosplit.so
  echo "== $lib: stamps" | tee -a $L
  VF_GEMM_9_STAGGER=100 timeout -k 10 200 python tools/gemm9_stamps.py --shapes 51200x2304x768 >> $L 2>&1 || { tail -20 $L; exit 1; }
  VF_GEMM_9_STAGGER=100 timeout -k 10 200 python tools/gemm9_stamps.py --shapes 51200x3072x768 --epi 1 >> $L 2>&1
  for epi in 0 1; do
    echo "== $lib: isolated, epi $epi, kinds 7 / 10" | tee -a $L
    timeout -k 10 300 python tools/bench_gemm.py --kind 7,10 --epi $epi >> $L 2>&1 || exit $?
  done
  echo "== $lib: forward" | tee -a $L
  for shape in xlmr-base xlmr-large; do
    echo "8p $shape" >> $L
    VF_GEMM_9=0 timeout -k 10 200 python tools/bench_rerank.py --shape $shape >> $L 2>&1 || exit $?
    echo "gemm9 $shape" >> $L
    VF_GEMM_9=1 timeout -k 10 200 python tools/bench_rerank.py --shape $shape >> $L 2>&1 || exit $?
  done
done
grep -E "^\{|^8p|^gemm9|==" $L | cut -c1-900
