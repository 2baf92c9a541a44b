#!/bin/bash
# split-K hand-over: the co-operative finish (this build) against the last-arrival form (veritasfi_amd/lib/libvf_head.so: the previous
# commit's transformer TU): parity tests, the products in isolation, then the forward at 13 / 100 pairs
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_skcoop.log
: > $L
R=$(pwd)
echo "== tests" >> $L
timeout -k 10 300 python -m pytest tests/test_gpu_encoder.py -q -m gpu -p no:cacheprovider -k "splitk or gemm_kernels_match" -x >> $L 2>&1 || { tail -30 $L; exit 1; }
S=6656x768x3072,6656x1024x4096,51200x768x3072,51200x1024x4096
for rep in 1 2; do
for lib in new head; do
  if [ $lib = head ]; then export VF_LIB_PATH=$R/veritasfi_amd/lib/libvf_head.so; else unset VF_LIB_PATH; fi
  for mode in 1 2; do
    echo "== lib $lib VF_SK_MODE=$mode epi 2" >> $L
    VF_SK_MODE=$mode timeout -k 10 200 python tools/bench_gemm.py --kind 7,0 --epi 2 --iters 50 --shapes $S >> $L 2>&1 || { tail -20 $L; exit 1; }
  done
done
done
unset VF_LIB_PATH
grep -E "^==|^\{|passed|failed" $L | cut -c1-130
