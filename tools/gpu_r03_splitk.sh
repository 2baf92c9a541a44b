#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x -p no:cacheprovider -k "splitk_tail or reranker_matches" -s > gpurun_out/pytest_splitk.log 2>&1; rc=$?
grep -a "split-K tail\|passed\|failed\|Error\|assert" gpurun_out/pytest_splitk.log | tail -12
if [ $rc -ne 0 ]; then exit $rc; fi
for env in "" "VF_NO_SPLITK_TAIL=1" ""; do
  for shape in xlmr-base xlmr-large; do
    echo "== [$env] $shape"
    env $env timeout -k 10 200 python tools/bench_rerank.py --shape $shape --iters 12 2>/dev/null | tail -1
  done
done
