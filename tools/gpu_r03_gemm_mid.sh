#!/bin/bash
# which product kernel for the per-rank batches of a data-parallel re-rank (13 / 25 / 50 pairs x 512 tokens)?
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_gemm_mid.log
for M in 6656 12800 25600; do
  for hk in "768 3072" "1024 4096"; do
    set -- $hk; H=$1; F=$2
    timeout -k 10 120 python3 tools/bench_gemm.py --shapes ${M}x$((3*H))x$H --kind 7,5,3 --epi 0 >> gpurun_out/r03_gemm_mid.log 2>&1 || exit 1
    timeout -k 10 120 python3 tools/bench_gemm.py --shapes ${M}x${H}x$H,${M}x${H}x$F --kind 7,5,3 --epi 2 >> gpurun_out/r03_gemm_mid.log 2>&1 || exit 1
    timeout -k 10 120 python3 tools/bench_gemm.py --shapes ${M}x${F}x$H --kind 7,5,3 --epi 1 >> gpurun_out/r03_gemm_mid.log 2>&1 || exit 1
  done
done
python3 - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r03_gemm_mid.log") if l.startswith("{")]
from collections import defaultdict
t = defaultdict(dict)
for r in rows:
    t[(r["shape"], r["epi"], r["vendor_lib_us"])][r["kind"]] = r["us"]
for (sh, epi, lib), d in t.items():
    print(f"{sh:18s} epi {epi}  8p {d.get(7)}  dma16 {d.get(5)}  128x128 {d.get(3)}  vendor(bias only) {lib}")
PY
