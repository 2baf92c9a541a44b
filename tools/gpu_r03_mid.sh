#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for pairs in 13 25; do
  rm -rf /tmp/prof_mid
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_mid -o out -- python3 $R/tools/bench_rerank.py --shape xlmr-base --pairs $pairs --iters 6 > $R/gpurun_out/mid_$pairs.log 2>/dev/null
  tail -1 $R/gpurun_out/mid_$pairs.log
  t=$(find /tmp/prof_mid -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_layer.py "$t" 12 | tee $R/gpurun_out/r03_rerank_layer_xlmr-base_${pairs}pairs.txt
done
