#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r05_legs4.log
: > $L
for v in "A=1" "GPU_MAX_HW_QUEUES=4"; do
  echo "== default line without the transformer legs: $v" >> $L
  env $v timeout -k 10 600 python bench.py --gpus 1 --no-rerank --no-cpu-baseline >> $L 2>/dev/null || exit 1
  echo "== standalone 1M: $v" >> $L
  env $v timeout -k 10 300 python bench.py --rows 1000000 --steps 200 --warmup 20 --no-rerank --no-cpu-baseline >> $L 2>/dev/null || exit 1
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_legs4.log"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        j = json.loads(l); print("   main", j["value"], j["ms_per_step"], j["roofline"]["frac"])
        for leg in ("c2", "shard8"):
            x = j.get(leg)
            if x: print("  ", leg, x.get("queries_per_s"), x.get("ms_per_step"), x.get("roofline", {}).get("frac"), x.get("error"))
PY
