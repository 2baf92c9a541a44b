#!/bin/bash
# round 5: split-K partials kept in the XCD's L2 (plain stores, relaxed arrival) against the write-through form of round 4 (VF_SK_DBG=64):
# parity tests, the sub-round long-K products, the forward at 13 / 25 / 100 pairs
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
L=$R/gpurun_out/r05_splitk_l2_partials.log
: > $L
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider -x -k "gemm9_whole_product or splitk_tail or two_handles or gemm_kernels_match" >> $L 2>&1; rc=$?
tail -3 $L
[ $rc -ne 0 ] && tail -60 $L && exit $rc
for rep in 1 2; do
for dbg in 0 64; do
  echo "== products, VF_SK_DBG=$dbg" >> $L
  VF_SK_DBG=$dbg timeout -k 10 300 python tools/bench_gemm.py --kind 0 --epi 2 --shapes 6656x768x3072,4096x768x3072,6656x1024x4096,12800x768x3072 >> $L 2>&1 || exit 1
  for pairs in 13 25 100; do
    echo "== forward xlmr-base pairs $pairs VF_SK_DBG=$dbg" >> $L
    VF_SK_DBG=$dbg timeout -k 10 300 python tools/bench_rerank.py --shape xlmr-base --pairs $pairs --iters 30 2>/dev/null | tail -1 >> $L || exit 1
  done
  echo "== forward xlmr-large pairs 13 VF_SK_DBG=$dbg" >> $L
  VF_SK_DBG=$dbg timeout -k 10 300 python tools/bench_rerank.py --shape xlmr-large --pairs 13 --iters 12 2>/dev/null | tail -1 >> $L || exit 1
done
done
grep -E "^==|^\{" $L | cut -c1-260
