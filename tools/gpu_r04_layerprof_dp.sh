#!/bin/bash
# per-layer kernel times of the re-rank forward at the per-rank batch sizes of a data-parallel re-rank (100 / 50 / 25 / 13 pairs)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
L=$R/gpurun_out/r04_layerprof_dp.log
: > $L
cd /tmp && export TMPDIR=/tmp
for pairs in "$@"; do
  rm -rf /tmp/prof_rr
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rr -o out -- python3 $R/tools/bench_rerank.py --shape xlmr-base --pairs $pairs --iters 4 > /tmp/rr.log 2>/dev/null
  echo "== pairs $pairs $(tail -1 /tmp/rr.log)" >> $L
  t=$(find /tmp/prof_rr -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_layer.py "$t" 12 >> $L
done
cat $L | cut -c1-200
