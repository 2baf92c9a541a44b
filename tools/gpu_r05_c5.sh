#!/bin/bash
# configs[4] at full size and at its 8-GPU shard, default kernel (k_scan_wide8); the wide tests first
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r05_c5.log
: > $L
timeout -k 10 400 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -p no:cacheprovider -x -k "wide or c5_shape or hostile or certificate" >> $L 2>&1; rc=$?
tail -3 $L
[ $rc -ne 0 ] && tail -40 $L && exit $rc
for rows in 1250000 10000000; do
  C5="--rows $rows --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --steps 10 --warmup 2"
  echo "== rows $rows" | tee -a $L
  timeout -k 10 400 python bench.py $C5 >> $L 2>gpurun_out/r05_c5.err || { tail -20 gpurun_out/r05_c5.err; exit 1; }
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_c5.log"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        j = json.loads(l); r = j["roofline"]
        print("  value", j["value"], "ms/step", j["ms_per_step"], "launch", r["avg_launch_ms"], "TF", r["achieved"], "frac", r["frac"], r["kernel"][:24], j.get("search_stats"))
PY
