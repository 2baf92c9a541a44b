#!/bin/bash
# after the sub-round split entered the dispatch: product + encoder tests, the forward at 13 / 25 / 100 pairs (both shapes)
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_skcoop_forward.log
: > $L
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -q -m gpu -p no:cacheprovider -x > gpurun_out/r04_skcoop_tests.log 2>&1 || { tail -30 gpurun_out/r04_skcoop_tests.log; exit 1; }
tail -2 gpurun_out/r04_skcoop_tests.log
for shape in xlmr-base xlmr-large; do
  for pairs in 13 25 100; do
    for sk in 1 0; do
      echo "== $shape pairs=$pairs VF_SPLITK_TAIL=$sk" >> $L
      VF_SPLITK_TAIL=$sk timeout -k 10 200 python tools/bench_rerank.py --shape $shape --pairs $pairs >> $L 2>&1 || { tail $L; exit 1; }
    done
  done
done
grep -E "^==|^\{" $L | cut -c1-200
