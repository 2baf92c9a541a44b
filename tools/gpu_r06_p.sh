#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_scan2r_1m_ab.log
: > $L
for rep in 1 2 3; do
  for impl in 4 5; do
      timeout -k 10 200 python3 bench.py --gpus 1 --rows 1000000 --steps 300 --warmup 30 --no-rerank --no-cpu-baseline --no-shard-legs --no-startup --opt scan_impl=$impl > gpurun_out/_ab.json 2>/dev/null || { echo fail; exit 1; }
      python3 - $rep $impl <<'PY' >> $L
import json, sys
j = json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1]); r = j["roofline"]
print(f"rep {sys.argv[1]} scan_impl {sys.argv[2]} rows 1000000: {j['ms_per_step']:.4f} ms/step  p50 {j['p50_ms_per_step']}  interval frac {r['frac']}  isolated {r.get('isolated_launch', {}).get('frac')}  cand/q {j['search_stats']['candidates_per_query']}")
PY
  done
done
cat $L
