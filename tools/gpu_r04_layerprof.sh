#!/bin/bash
# per-layer kernel times of the re-rank forward, 8-phase products vs k_gemm9_tn (rocprofv3 kernel trace), + gemm9 stamps at the large shapes
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
L=$R/gpurun_out/r04_layerprof.log
: > $L



cd /tmp && export TMPDIR=/tmp
for shape in xlmr-base xlmr-large; do
  layers=12; [ $shape = xlmr-large ] && layers=24
  for v in 0 1; do
    export VF_GEMM_9=$v
    rm -rf /tmp/prof_rr
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rr -o out -- python3 $R/tools/bench_rerank.py --shape $shape --iters 4 > /tmp/rr.log 2>/dev/null
    echo "== $shape VF_GEMM_9=$v $(tail -1 /tmp/rr.log)" >> $L
    t=$(find /tmp/prof_rr -name "*kernel_trace.csv" | head -1)
    python3 $R/tools/trace_layer.py "$t" $layers >> $L
  done
done
cat $L | cut -c1-1100
