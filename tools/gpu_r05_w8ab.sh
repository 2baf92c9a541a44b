#!/bin/bash
# round 5: k_scan_wide8 as two 4-wave workgroups per CU (wide8_waves=4) against the one 8-wave workgroup (=8); the wide tests first
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r05_w8ab.log
: > $L
timeout -k 10 500 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -p no:cacheprovider -x -k "wide or c5_shape or hostile or certificate" >> $L 2>&1; rc=$?
tail -3 $L
[ $rc -ne 0 ] && tail -60 $L && exit $rc
for rows in 1250000 10000000; do
  for w in 4 8; do
    C5="--rows $rows --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --steps 10 --warmup 2 --opt wide8_waves=$w"
    echo "== rows $rows wide8_waves=$w" | tee -a $L
    timeout -k 10 300 python bench.py $C5 >> $L 2>gpurun_out/r05_w8ab.err || { tail -20 gpurun_out/r05_w8ab.err; exit 1; }
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_w8ab.log"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        j = json.loads(l); r = j["roofline"]
        print("  value", j["value"], "ms/step", j["ms_per_step"], "launch", r["avg_launch_ms"], "TF", r["achieved"], "frac", r["frac"], r["kernel"][:24], j.get("search_stats"))
PY
