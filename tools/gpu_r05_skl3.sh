#!/bin/bash
# round 5: the K cut after the relaxed arrival: tests; the products of a 13-pair layer; the same with the cut allowed at K = 768 (out projection)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
L=$R/gpurun_out/r05_splitk_relaxed.log
: > $L
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider -x -k "gemm9_whole_product or splitk_tail or two_handles or gemm_kernels_match" >> $L 2>&1; rc=$?
tail -3 $L
[ $rc -ne 0 ] && tail -60 $L && exit $rc
SH=6656x768x3072,6656x768x768,6656x2304x768,6656x3072x768,12800x768x768,4096x768x768
for rep in 1 2; do
  echo "== products, default" >> $L
  timeout -k 10 300 python tools/bench_gemm.py --kind 0 --epi 2 --shapes $SH >> $L 2>&1 || exit 1
  echo "== products, VF_GEMM_9_SPLIT_MINK=768 VF_GEMM_9_SPLIT_MINKT=4" >> $L
  VF_GEMM_9_SPLIT_MINK=768 VF_GEMM_9_SPLIT_MINKT=4 timeout -k 10 300 python tools/bench_gemm.py --kind 0 --epi 2 --shapes $SH >> $L 2>&1 || exit 1
  echo "== products, VF_GEMM_9_SPLIT_MINK=768 VF_GEMM_9_SPLIT_MINKT=6" >> $L
  VF_GEMM_9_SPLIT_MINK=768 VF_GEMM_9_SPLIT_MINKT=6 timeout -k 10 300 python tools/bench_gemm.py --kind 0 --epi 2 --shapes $SH >> $L 2>&1 || exit 1
  for env in "" "VF_GEMM_9_SPLIT_MINK=768 VF_GEMM_9_SPLIT_MINKT=4" "VF_GEMM_9_SPLIT_MINK=768 VF_GEMM_9_SPLIT_MINKT=6"; do
    for pairs in 13 8; do
      echo "== forward xlmr-base pairs $pairs $env" >> $L
      env $env timeout -k 10 300 python tools/bench_rerank.py --shape xlmr-base --pairs $pairs --iters 30 2>/dev/null | tail -1 >> $L || exit 1
    done
  done
done
grep -E "^==|^\{" $L | cut -c1-200
