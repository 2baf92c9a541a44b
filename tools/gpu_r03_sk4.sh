#!/bin/bash
# the forward with the split-K of partial rounds off / default (long-K tails in two) / general
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_sk4.log
for shape in xlmr-base xlmr-large; do
 for pairs in 13 25 50 100; do
  for mode in 0 1 2; do
    echo "== $shape pairs=$pairs mode=$mode: $(VF_SPLITK_TAIL=$mode timeout -k 10 200 python3 tools/bench_rerank.py --shape $shape --pairs $pairs --iters 12 2>/dev/null | tail -1)" >> gpurun_out/r03_sk4.log
  done
 done
done
cut -c1-150 gpurun_out/r03_sk4.log
