#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_c5_depth.log
: > $L
for rep in 1 2; do
for depth in 2 3 4; do
  VF_BENCH_DEPTH=$depth timeout -k 10 300 python3 bench.py --rows 10000000 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --no-startup --steps 24 --warmup 4 > gpurun_out/_c5.json 2>/dev/null || { echo fail; exit 1; }
  python3 - $depth <<'PY' >> $L
import json, sys
j = json.loads(open("gpurun_out/_c5.json").read().strip().splitlines()[-1]); r = j["roofline"]
print(f"configs[4] 10M x 1024 e4m3, batch 1024, top-1000, batches in flight {sys.argv[1]}: {j['value']:.1f} q/s  {j['ms_per_step']:.3f} ms/step  p50 {j['p50_ms_per_step']}  launch {r['avg_launch_ms']} ms  frac {r['frac']}")
PY
done
done
cat $L
