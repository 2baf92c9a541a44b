#!/bin/bash
# per-layer kernel times of the re-rank forward at the per-rank batch sizes of a 2/4/8-GPU data-parallel re-rank (50/25/13 pairs)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for shape in xlmr-base xlmr-large; do
 for pairs in 13 25 50; do
  rm -rf /tmp/prof_rr
  layers=12; [ $shape = xlmr-large ] && layers=24
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rr -o out -- python3 $R/tools/bench_rerank.py --shape $shape --pairs $pairs --iters 6 > $R/gpurun_out/rrmid_${shape}_$pairs.log 2>/dev/null
  echo "== $shape pairs=$pairs: $(tail -1 $R/gpurun_out/rrmid_${shape}_$pairs.log)" | tee -a $R/gpurun_out/r03_rerank_mid_layers.txt
  t=$(find /tmp/prof_rr -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_layer.py "$t" $layers | tee -a $R/gpurun_out/r03_rerank_mid_layers.txt
 done
done
