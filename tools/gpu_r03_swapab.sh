#!/bin/bash
# same-process-tree A/B of the 8-byte-write epilogue (shipped) against the two-byte-write one (lib/libvf_noswap.so: -DVF_8P_NO_SWAP)
mkdir -p gpurun_out
: > gpurun_out/r03_swapab.log
for rep in 1 2; do for lib in "" libvf_noswap.so; do
  L=""; [ -n "$lib" ] && L="$PWD/veritasfi_amd/lib/$lib"
  for shape in xlmr-base xlmr-large; do echo "lib=${lib:-shipped} $shape $(VF_LIB_PATH=$L timeout -k 10 200 python3 tools/bench_rerank.py --shape $shape --iters 12 2>/dev/null | tail -1 | cut -c50-120)" >> gpurun_out/r03_swapab.log; done
  for e in 0 1 2; do VF_LIB_PATH=$L timeout -k 10 200 python3 tools/bench_gemm.py --kind 7 --epi $e --check 0 --shapes 51200x2304x768,51200x1024x4096,51200x4096x1024 2>/dev/null | cut -c1-90 | sed "s/^/lib=${lib:-shipped} /" >> gpurun_out/r03_swapab.log; done
done; done
cat gpurun_out/r03_swapab.log
