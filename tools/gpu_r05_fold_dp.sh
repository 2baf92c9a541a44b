#!/bin/bash
# LayerNorm fold (round 3, off by default) at the data-parallel row counts, where the two LayerNorm launches are 10 % of a layer
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
L=gpurun_out/r05_ln_fold_dp.log
: > $L
for pairs in 13 25 50; do
  for rep in 1 2; do
    echo "== pairs $pairs default" >> $L
    timeout -k 10 120 python tools/bench_rerank.py --shape xlmr-base --pairs $pairs --iters 40 2>/dev/null | tail -1 >> $L || exit 1
    echo "== pairs $pairs VF_LN_FOLD=1 VF_GEMM_8P_MIN_WGS=0" >> $L
    VF_LN_FOLD=1 VF_GEMM_8P_MIN_WGS=0 timeout -k 10 120 python tools/bench_rerank.py --shape xlmr-base --pairs $pairs --iters 40 2>/dev/null | tail -1 >> $L || exit 1
    echo "== pairs $pairs VF_GEMM_8P_MIN_WGS=0 (8-phase kernel, no fold)" >> $L
    VF_GEMM_8P_MIN_WGS=0 timeout -k 10 120 python tools/bench_rerank.py --shape xlmr-base --pairs $pairs --iters 40 2>/dev/null | tail -1 >> $L || exit 1
  done
done
cut -c1-220 $L
