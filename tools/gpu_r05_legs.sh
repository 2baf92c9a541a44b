#!/bin/bash
# are the c2 / shard8 legs of the default line slower than the same workloads run on their own?  same box, same binary
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r05_legs.log
: > $L
echo "== default line without the transformer legs" >> $L
timeout -k 10 600 python bench.py --gpus 1 --no-rerank --no-cpu-baseline >> $L 2>/dev/null || exit 1
echo "== standalone 1M" >> $L
timeout -k 10 300 python bench.py --rows 1000000 --steps 200 --warmup 20 --no-rerank --no-cpu-baseline >> $L 2>/dev/null || exit 1
echo "== standalone 1.25M with the exchange" >> $L
VF_BENCH_LAUNCH=1 VF_BENCH_FORCE_EXCHANGE=1 timeout -k 10 300 python bench.py --gpus 1 --rows 1250000 --steps 200 --warmup 20 --verify --no-rerank --no-cpu-baseline >> $L 2>/dev/null || exit 1
python - <<'PY'
import json
for l in open("gpurun_out/r05_legs.log"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        j = json.loads(l)
        print("  main", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("avg_launch_ms"), j["search_stats"])
        for leg in ("c2", "shard8"):
            x = j.get(leg)
            if x: print("  ", leg, x.get("queries_per_s"), x.get("ms_per_step"), x.get("roofline", {}).get("frac"), x.get("roofline", {}).get("avg_launch_ms"), x.get("search_stats"), x.get("error"))
PY
