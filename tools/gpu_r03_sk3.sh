#!/bin/bash
# where does the split-K hand-over spend its time?  dbg 1 = no partial stores, 2 = no read-back, 3 = neither (timing only: results wrong)
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_sk3.log
for dbg in 0 1 2 3; do
  VF_SK_MODE=2 VF_SK_DBG=$dbg timeout -k 10 120 python3 tools/bench_gemm.py --check 0 --shapes 6656x768x768,6656x768x3072,51200x768x768,51200x1024x4096 --kind 7 --epi 2 >> gpurun_out/r03_sk3.log 2>&1 || exit 1
done
VF_SK_MODE=0 timeout -k 10 120 python3 tools/bench_gemm.py --check 0 --shapes 6656x768x768,6656x768x3072,51200x768x768,51200x1024x4096 --kind 7 --epi 2 >> gpurun_out/r03_sk3.log 2>&1
grep -a "^{" gpurun_out/r03_sk3.log | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['shape'], 'mode', d['sk_mode'], 'us', d['us'], 'readbacks l2/mem', d['sk_readbacks_l2_mem'])
"
