#!/bin/bash
# sustained (in-forward) A/B of the MFMA issue order inside a quadrant of k_gemm8p_tn: shipped = consecutive MFMAs share the A
# fragment (t outer, u inner), lib/libvf_tinner.so = they share the W fragment (-DVF_8P_QUAD_T_INNER)
mkdir -p gpurun_out
: > gpurun_out/r03_orderab.log
for rep in 1 2; do for lib in "" libvf_tinner.so; do
  L=""; [ -n "$lib" ] && L="$PWD/veritasfi_amd/lib/$lib"
  for shape in xlmr-base xlmr-large; do echo "lib=${lib:-shipped} $shape $(VF_LIB_PATH=$L timeout -k 10 200 python3 tools/bench_rerank.py --shape $shape --iters 12 2>/dev/null | tail -1 | cut -c50-120)" >> gpurun_out/r03_orderab.log; done
done; done
cat gpurun_out/r03_orderab.log
