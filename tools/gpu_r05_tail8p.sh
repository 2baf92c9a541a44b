#!/bin/bash
# round 5: after the relaxed arrival, the 8-phase kernel's split-K tail against the persistent kernel on long-K products of 1-3 rounds
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
L=$R/gpurun_out/r05_tail8p_vs_gemm9.log
: > $L
SH=25600x768x3072,38400x768x3072,51200x768x3072,19200x768x3072,25600x1024x4096,12800x1024x4096
for rep in 1 2; do
  for kind in 0 7 10; do
    echo "== products epi 2 kind $kind" >> $L
    timeout -k 10 300 python tools/bench_gemm.py --kind $kind --epi 2 --shapes $SH >> $L 2>&1 || exit 1
  done
done
grep -E "^==|^\{" $L | cut -c1-200
