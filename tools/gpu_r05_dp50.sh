#!/bin/bash
# round 5: the 50-pair forward (one rank's share at 2 GPUs) with FFN-down on the 8-phase kernel's split-K tail (new rule) against the persistent kernel
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
L=$R/gpurun_out/r05_dp50_tail_rule.log
: > $L
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider -x -k "stream_k or gemm9_whole_product or splitk_tail or gemm_kernels_match or rank_order" >> $L 2>&1; rc=$?
tail -2 $L
[ $rc -ne 0 ] && tail -60 $L && exit $rc
for rep in 1 2 3; do
  for env in "" "VF_SPLITK_TAIL=0"; do
    for pairs in 50 100; do
      echo "== forward xlmr-base pairs $pairs $env" >> $L
      env $env timeout -k 10 300 python tools/bench_rerank.py --shape xlmr-base --pairs $pairs --iters 30 2>/dev/null | tail -1 >> $L || exit 1
    done
  done
done
for pairs in 13 25; do
  echo "== forward xlmr-base pairs $pairs" >> $L
  timeout -k 10 300 python tools/bench_rerank.py --shape xlmr-base --pairs $pairs --iters 30 2>/dev/null | tail -1 >> $L || exit 1
done
grep -E "^==|^\{" $L | cut -c1-160
