#!/bin/bash
# round 5: stream-K in the persistent product kernel: parity, then products and forwards with it on / off in one call
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
L=$R/gpurun_out/r05_streamk.log
: > $L
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider -x -s -k "stream_k or gemm9_whole_product or splitk_tail or gemm_kernels_match" >> $L 2>&1; rc=$?
grep -E "gemm9 stream-K|passed|failed" $L | cut -c1-200
[ $rc -ne 0 ] && tail -60 $L && exit $rc
[ "$1" = "parity" ] && exit 0
SH=6656x3072x768,12800x3072x768,12800x768x3072,12800x2304x768,25600x768x3072,25600x3072x768,51200x768x3072,51200x3072x768
for rep in 1 2; do
  for sk in 1 0; do
    for epi in 1 2; do
      echo "== products epi $epi VF_GEMM_9_STREAMK=$sk" >> $L
      VF_GEMM_9_STREAMK=$sk timeout -k 10 300 python tools/bench_gemm.py --kind 0 --epi $epi --shapes $SH >> $L 2>&1 || exit 1
    done
    for pairs in 13 25 50 100; do
      echo "== forward xlmr-base pairs $pairs VF_GEMM_9_STREAMK=$sk" >> $L
      VF_GEMM_9_STREAMK=$sk timeout -k 10 300 python tools/bench_rerank.py --shape xlmr-base --pairs $pairs --iters 30 2>/dev/null | tail -1 >> $L || exit 1
    done
  done
done
grep -E "^==|^\{" $L | cut -c1-230
