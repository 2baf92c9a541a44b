#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate $R/tools/ubench/mfma_rate.hip 2>/dev/null && timeout -k 5 120 /tmp/mfma_rate > $R/gpurun_out/mfma_rate.log 2>&1
cat $R/gpurun_out/mfma_rate.log
cd /tmp && export TMPDIR=/tmp
for shape in xlmr-base xlmr-large; do
  rm -rf /tmp/prof_rr
  layers=12; [ $shape = xlmr-large ] && layers=24
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rr -o out -- python3 $R/tools/bench_rerank.py --shape $shape --iters 4 > $R/gpurun_out/rr_$shape.log 2>/dev/null
  tail -1 $R/gpurun_out/rr_$shape.log
  t=$(find /tmp/prof_rr -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_layer.py "$t" $layers | tee $R/gpurun_out/r03_rerank_layer_$shape.txt
  f=$(find /tmp/prof_rr -name "*kernel_stats.csv" | head -1)
  head -12 "$f" | cut -c1-200 > $R/gpurun_out/r03_kernel_stats_rerank_${shape}_100x512.csv
done
