#!/bin/bash
# round 5: half items in k_attention2 (partial last round / small batches): parity, then the forward with them on / off in one call
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
L=$R/gpurun_out/r05_attention_half_items.log
: > $L
timeout -k 10 700 python -m pytest tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider -x -k "attention" >> $L 2>&1; rc=$?
tail -2 $L
[ $rc -ne 0 ] && tail -60 $L && exit $rc
for rep in 1 2 3; do
  for on in 1 0; do
    for pairs in 25 50 8 4 100 13; do
      echo "== forward xlmr-base pairs $pairs VF_ATT_HALVES=$on" >> $L
      VF_ATT_HALVES=$on timeout -k 10 300 python tools/bench_rerank.py --shape xlmr-base --pairs $pairs --iters 30 2>/dev/null | tail -1 >> $L || exit 1
    done
  done
done
grep -E "^==|^\{" $L | cut -c1-140
