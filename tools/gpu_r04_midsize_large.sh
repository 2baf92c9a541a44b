#!/bin/bash
# forced-kernel sweep at the 1024 / 4096 layer shapes, mid-size row counts
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_midsize_large.log
: > $L
for M in 6656 12800 25600; do
  for epi in 0 2; do
    shapes="${M}x3072x1024,${M}x1024x1024,${M}x4096x1024,${M}x1024x4096"
    [ $epi = 2 ] && shapes="${M}x1024x1024,${M}x1024x4096"
    echo "== M $M epi $epi" >> $L
    VF_SK_MODE=1 timeout -k 10 300 python tools/bench_gemm.py --kind 0,3,5,7,10 --epi $epi --shapes $shapes >> $L 2>&1 || exit 1
  done
done
python3 - <<'PY'
import json, collections
rows=collections.defaultdict(dict)
for l in open('gpurun_out/r04_midsize_large.log'):
    if l.startswith('{'):
        j=json.loads(l); rows[(j['shape'], j['epi'])][j['kind']]=(j['us'], j['vendor_lib_us'])
print("shape epi | auto  k3  k5  k7  k10 | vendor | best")
for key in sorted(rows, key=lambda k:(int(k[0].split('x')[0]), k[1], k[0])):
    r=rows[key]; v=r[0][1]; best=min((r[k][0],k) for k in r if k!=0)
    print(f"{key[0]:16s} {key[1]} | " + " ".join(f"{r[k][0]:6.1f}" for k in (0,3,5,7,10)) + f" | {v:6.1f} | k{best[1]} {best[0]/v:.2f}x  auto {r[0][0]/v:.2f}x")
PY
