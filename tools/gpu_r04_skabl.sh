#!/bin/bash
# where the time of a product cut whole goes: partial stores off (1), read-backs off (2), both (3) -- timing only
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_skabl.log
: > $L
for dbg in 0 1 2 3; do
  echo "== VF_SK_DBG=$dbg (mode 2)" >> $L
  VF_SK_MODE=2 VF_SK_DBG=$dbg timeout -k 10 100 python tools/bench_gemm.py --kind 7 --epi 2 --check 0 --iters 50 --shapes 6656x768x3072,6656x1024x4096,6656x768x768 >> $L 2>&1 || { tail $L; exit 1; }
done
echo "== no cut" >> $L
VF_SK_MODE=0 timeout -k 10 100 python tools/bench_gemm.py --kind 7 --epi 2 --check 0 --iters 50 --shapes 6656x768x3072,6656x1024x4096,6656x768x768,6656x768x1024 >> $L 2>&1
grep -E "^==|^\{" $L | cut -c1-100
