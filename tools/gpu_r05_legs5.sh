#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --gpus 1 --no-rerank --no-cpu-baseline > gpurun_out/r05_legs5.log 2>gpurun_out/r05_legs5.err || { tail -20 gpurun_out/r05_legs5.err; exit 1; }
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05_legs5.log").read().strip().splitlines()[-1])
print("main", j["value"], j["ms_per_step"], j["roofline"]["frac"])
for leg in ("c2", "shard8"):
    x = j.get(leg) or {}
    print(leg, {k: x.get(k) for k in ("queries_per_s", "ms_per_step", "child_wall_s", "error", "command")}, (x.get("roofline") or {}).get("frac"), (x.get("roofline") or {}).get("traffic"), x.get("exchange"))
PY
