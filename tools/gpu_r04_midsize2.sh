#!/bin/bash
# after the mid-size dispatch change: GEMM + encoder tests, the forward at 100 / 50 / 25 / 13 pairs, the vendor chain at mid sizes
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_midsize2.log
: > $L
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py tests/test_vision.py -m gpu -q -p no:cacheprovider -x >> $L 2>&1; rc=$?
tail -3 $L
[ $rc -ne 0 ] && tail -40 $L && exit $rc
for shape in xlmr-base xlmr-large; do
  for p in 100 50 25 13; do
    echo "== $shape pairs $p" >> $L
    timeout -k 10 200 python tools/bench_rerank.py --shape $shape --pairs $p 2>/dev/null | grep "^{" >> $L || exit 1
  done
done
timeout -k 10 200 python tools/bench_gemm_chain.py >> $L 2>&1
grep -E "^==|^\{" $L | cut -c1-260
