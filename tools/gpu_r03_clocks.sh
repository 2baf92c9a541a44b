#!/bin/bash
# engine clock and power while a product kernel runs back to back (rocm-smi polled beside the loop): what does "peak" assume?
mkdir -p gpurun_out
: > gpurun_out/r03_clocks.log
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk" | tr -s ' ' | cut -c1-110 >> gpurun_out/r03_clocks.log; echo "--" >> gpurun_out/r03_clocks.log; sleep 0.5; done ) &
POLL=$!
sleep 2
echo "## idle above; now k_gemm8p_tn 51200x768x3072 x 4000" >> gpurun_out/r03_clocks.log
timeout -k 10 120 python3 tools/bench_gemm.py --kind 7 --epi 0 --check 0 --iters 4000 --shapes 51200x768x3072 2>/dev/null | cut -c1-120 >> gpurun_out/r03_clocks.log
echo "## now the register-only MFMA loop" >> gpurun_out/r03_clocks.log
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/ubench/mfma_rate.hip 2>/dev/null && for i in 1 2 3 4 5 6; do timeout -k 5 60 /tmp/mfma_rate | head -3 >> gpurun_out/r03_clocks.log; done
wait $POLL
cat gpurun_out/r03_clocks.log | head -150
