#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r05_pretrained.log
timeout -k 10 900 python -m pytest tests/test_pretrained.py tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider -x -k "pretrained or from_config or pair_inputs or rank_order or reranker_matches or pooling_variants" > $L 2>&1; rc=$?
tail -5 $L
[ $rc -ne 0 ] && tail -70 $L
grep -E "from_pretrained|LLM re-ranker|cross-encoder|xlmr-" $L | head -20
exit $rc
