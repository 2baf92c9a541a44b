#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_encoder.py tests/test_vision.py -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_swap.log 2>&1; rc=$?
tail -3 gpurun_out/pytest_swap.log
if [ $rc -ne 0 ]; then grep -a "Error\|assert" gpurun_out/pytest_swap.log | head; exit $rc; fi
: > gpurun_out/r03_swapepi.log
for e in 0 1 2; do timeout -k 10 200 python3 tools/bench_gemm.py --kind 7 --epi $e --check 0 2>/dev/null | cut -c1-100 >> gpurun_out/r03_swapepi.log; done
for shape in xlmr-base xlmr-large; do echo "$shape $(timeout -k 10 200 python3 tools/bench_rerank.py --shape $shape --iters 12 2>/dev/null | tail -1 | cut -c50-130)" >> gpurun_out/r03_swapepi.log; done
timeout -k 10 300 python3 tools/bench_gemm_chain.py --rows 51200 2>/dev/null | cut -c1-230 >> gpurun_out/r03_swapepi.log
timeout -k 10 300 python3 tools/bench_gemm_chain.py --rows 51200 --forward-epilogues 2>/dev/null | cut -c1-230 >> gpurun_out/r03_swapepi.log
cat gpurun_out/r03_swapepi.log
