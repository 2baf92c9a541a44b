#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r05_legs2.log
: > $L
for v in "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=16" "GPU_MAX_HW_QUEUES=2"; do
  echo "== default line without the transformer legs: $v" >> $L
  env $v timeout -k 10 600 python bench.py --gpus 1 --no-rerank --no-cpu-baseline --steps 50 --warmup 5 >> $L 2>/dev/null || exit 1
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_legs2.log"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        j = json.loads(l)
        for leg in ("c2", "shard8"):
            x = j.get(leg)
            if x: print("  ", leg, x.get("queries_per_s"), x.get("ms_per_step"), x.get("roofline", {}).get("frac"), x.get("error"))
PY
