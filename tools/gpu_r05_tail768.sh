#!/bin/bash
# round 5: after the relaxed arrival, the split-K TAIL on K = 768 products of 1.2 / 2.3 rounds (8-phase kernel, VF_SPLITK_TAIL=2 = the general form) against the default
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
L=$R/gpurun_out/r05_tail_split_k768.log
: > $L
SH=6656x3072x768,12800x3072x768,12800x2304x768,25600x3072x768,6656x2304x768
for rep in 1 2; do
  echo "== default (kind 0)" >> $L
  timeout -k 10 300 python tools/bench_gemm.py --kind 0 --epi 1 --shapes $SH >> $L 2>&1 || exit 1
  echo "== 8-phase kernel, no tail split (kind 7, VF_SPLITK_TAIL=0)" >> $L
  VF_SPLITK_TAIL=0 timeout -k 10 300 python tools/bench_gemm.py --kind 7 --epi 1 --shapes $SH >> $L 2>&1 || exit 1
  echo "== 8-phase kernel, general tail split (kind 7, VF_SPLITK_TAIL=2)" >> $L
  VF_SPLITK_TAIL=2 timeout -k 10 300 python tools/bench_gemm.py --kind 7 --epi 1 --shapes $SH >> $L 2>&1 || exit 1
done
grep -E "^==|^\{" $L | cut -c1-220
