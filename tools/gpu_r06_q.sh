#!/bin/bash
# e4m3 rows at batch 64 (k_scan): whole chip + ordered scans (the rule above 6M rows) against CU split + overlapping scans
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_fp8_narrow_split.log
: > $L
for rep in 1 2; do
  for dim in 768 1024; do
    for o in "" "--opt aux_cus=32 --opt overlap_scans=1"; do
      timeout -k 10 300 python3 bench.py --gpus 1 --rows 10000000 --dim $dim --corpus-dtype fp8 --steps 40 --warmup 8 --no-rerank --no-cpu-baseline --no-shard-legs --no-startup $o > gpurun_out/_ab.json 2>/dev/null || { echo fail; exit 1; }
      python3 - $rep $dim "$o" <<'PY' >> $L
import json, sys
j = json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1]); r = j["roofline"]
print(f"rep {sys.argv[1]} 10M x {sys.argv[2]} e4m3, batch 64 [{sys.argv[3] or 'default'}]: {j['ms_per_step']:.4f} ms/step  frac {r['frac']}  isolated {r.get('isolated_launch', {}).get('frac')}  kernel {r['kernel'][:24]}")
PY
    done
  done
done
cat $L
