#!/bin/bash
# the driver's command, whole line
set -o pipefail
mkdir -p gpurun_out
t0=$(date +%s)
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_cmd.log 2> gpurun_out/r06_bench_driver_cmd.err || { tail -30 gpurun_out/r06_bench_driver_cmd.err; exit 1; }
echo "wall $(( $(date +%s) - t0 )) s"
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/r06_bench_driver_cmd.log").read().strip().splitlines()[-1])
print("main", j["value"], j["ms_per_step"], j["roofline"]["frac"], "verified", j.get("verified"))
for leg in ("c2", "shard8"):
    x = j.get(leg) or {}
    print(leg, {k: x.get(k) for k in ("queries_per_s", "ms_per_step", "child_wall_s", "error")}, (x.get("roofline") or {}).get("frac"), ((x.get("roofline") or {}).get("isolated_launch") or {}).get("frac"), (x.get("exchange") or {}).get("merged_equals_direct"))
print("rerank_p50_ms", j.get("rerank_p50_ms"), "dp", (j.get("rerank") or {}).get("dp_share"))
print("c4", json.dumps((j.get("c4") or {}).get("p50_ms")), (j.get("c4") or {}).get("retriever_build_s"), (j.get("c4") or {}).get("error"))
print("c5", {k: (j.get("c5") or {}).get(k) for k in ("queries_per_s", "ms_per_step", "error")})
print("embed", j.get("embed"))
print("embed_texts", json.dumps(j.get("embed_texts")))
print("rerank_texts", json.dumps(j.get("rerank_texts")))
print("startup", json.dumps(j.get("startup")))
PY
