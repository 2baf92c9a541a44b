#!/bin/bash
# round 5: the whole-product K cut inside the persistent product kernel: (1) parity tests; (2) the products of a 13-pair layer, cut on / off;
# (3) the re-rank forward at the per-rank batch sizes of a data-parallel re-rank, cut on / off; (4) per-layer kernel times
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
L=$R/gpurun_out/r05_split9.log
: > $L
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider -x -k "gemm9_whole_product or splitk_tail or two_handles or gemm_kernels_match" >> $L 2>&1; rc=$?
tail -3 $L
[ $rc -ne 0 ] && tail -60 $L && exit $rc
for split in 1 0; do
  echo "== products, VF_GEMM_9_SPLIT=$split" >> $L
  for epi in 0 2; do
    VF_GEMM_9_SPLIT=$split timeout -k 10 300 python tools/bench_gemm.py --kind 0 --epi $epi --shapes 6656x768x3072,4096x768x3072,6656x1024x4096 >> $L 2>&1 || exit 1
  done
  for pairs in 13 25 100; do
    echo "== forward xlmr-base pairs $pairs VF_GEMM_9_SPLIT=$split" >> $L
    VF_GEMM_9_SPLIT=$split timeout -k 10 300 python tools/bench_rerank.py --shape xlmr-base --pairs $pairs --iters 12 2>/dev/null | tail -1 >> $L || exit 1
  done
  echo "== forward xlmr-large pairs 13 VF_GEMM_9_SPLIT=$split" >> $L
  VF_GEMM_9_SPLIT=$split timeout -k 10 300 python tools/bench_rerank.py --shape xlmr-large --pairs 13 --iters 8 2>/dev/null | tail -1 >> $L || exit 1
done
cd /tmp && export TMPDIR=/tmp
for pairs in 13; do
  rm -rf /tmp/prof_rr
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rr -o out -- python3 $R/tools/bench_rerank.py --shape xlmr-base --pairs $pairs --iters 4 > /tmp/rr.log 2>/dev/null
  echo "== layer profile, pairs $pairs $(tail -1 /tmp/rr.log)" >> $L
  t=$(find /tmp/prof_rr -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_layer.py "$t" 12 >> $L
done
grep -E "^==|^\{|^ +[0-9]+ |sum of" $L | cut -c1-330
