#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r05_legs3.log
: > $L
for v in "A=1" "VF_BENCH_BALLAST_GB=16" "VF_BENCH_BALLAST_GB=64" "A=2"; do
  echo "== standalone 1M: $v" >> $L
  env $v timeout -k 10 300 python bench.py --rows 1000000 --steps 200 --warmup 20 --no-rerank --no-cpu-baseline >> $L 2>/dev/null || exit 1
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_legs3.log"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        j = json.loads(l); print("  ", j["value"], j["ms_per_step"], j["roofline"]["frac"])
PY
