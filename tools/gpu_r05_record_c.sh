#!/bin/bash
# round 5 record, part C: the artefacts VERDICT r04 item 1 names — the layer's four products as a chain against the vendor library at the
# data-parallel row counts, and the re-rank forward with its per-layer kernel times at 100 / 50 / 25 / 13 pairs
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
L=$R/gpurun_out/r05_gemm_chain_vendor_vs_repo.log
: > $L
timeout -k 10 200 python tools/bench_gemm_chain.py --rows 6656,12800,25600,51200 2>/dev/null >> $L || exit 1
timeout -k 10 200 python tools/bench_gemm_chain.py --rows 6656,12800 --forward-epilogues 2>/dev/null >> $L || exit 1
timeout -k 10 200 python tools/bench_gemm_chain.py --hidden 1024 --ffn 4096 --rows 6656,12800,25600,51200 2>/dev/null >> $L || exit 1
cat $L | cut -c1-260
P=$R/gpurun_out/r05_rerank_layer_dp_batches.txt
: > $P
cd /tmp && export TMPDIR=/tmp
for pairs in 100 50 25 13; do
  timeout -k 10 120 python3 $R/tools/bench_rerank.py --shape xlmr-base --pairs $pairs --iters 30 2>/dev/null | tail -1 > /tmp/rr_plain.log || exit 1
  rm -rf /tmp/prof_rr
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rr -o out -- python3 $R/tools/bench_rerank.py --shape xlmr-base --pairs $pairs --iters 4 > /tmp/rr.log 2>/dev/null || exit 1
  echo "== pairs $pairs (unprofiled, 30 iterations) $(cat /tmp/rr_plain.log)" >> $P
  t=$(find /tmp/prof_rr -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_layer.py "$t" 12 >> $P
done
cut -c1-200 $P
