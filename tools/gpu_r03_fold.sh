#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x -p no:cacheprovider -k "folded or embedding_encoder or reranker_matches or xlmr_large or ragged_batch" -s > gpurun_out/pytest_fold.log 2>&1; rc=$?
grep -a "ln-fold\|hidden mean\|passed\|failed\|Error\|assert" gpurun_out/pytest_fold.log | tail -30
if [ $rc -ne 0 ]; then exit $rc; fi
for env in "" "VF_NO_LN_FOLD=1"; do
  for shape in xlmr-base xlmr-large; do
    echo "== $env $shape"
    env $env timeout -k 10 200 python tools/bench_rerank.py --shape $shape --iters 12 2>/dev/null | tail -1
  done
done
