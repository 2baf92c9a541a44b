#!/bin/bash
# k_gemm8p_tn with two phases of 32 MFMAs per K-tile (VF_GEMM_8P_LOOP2=1) against the four-phase loop: parity, products, forward
set -o pipefail
mkdir -p gpurun_out
VF_GEMM_8P_LOOP2=1 timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x -p no:cacheprovider -k "gemm or reranker_matches or xlmr_large or embedding_encoder or splitk or layernorm_folded or gemma or qwen" > gpurun_out/pytest_loop2.log 2>&1; rc=$?
tail -3 gpurun_out/pytest_loop2.log
if [ $rc -ne 0 ]; then grep -a "Error\|assert" gpurun_out/pytest_loop2.log | head; exit $rc; fi
: > gpurun_out/r03_loop2.log
for l in 0 1 0 1; do
  echo "== loop2=$l" >> gpurun_out/r03_loop2.log
  VF_GEMM_8P_LOOP2=$l timeout -k 10 200 python3 tools/bench_gemm.py --kind 7 --epi 0 --check 0 2>/dev/null | cut -c1-100 >> gpurun_out/r03_loop2.log
done
for shape in xlmr-base xlmr-large; do for l in 0 1 0 1; do echo "$shape loop2=$l $(VF_GEMM_8P_LOOP2=$l timeout -k 10 200 python3 tools/bench_rerank.py --shape $shape --iters 12 2>/dev/null | tail -1 | cut -c50-130)" >> gpurun_out/r03_loop2.log; done; done
cat gpurun_out/r03_loop2.log
