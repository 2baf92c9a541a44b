#!/bin/bash
# round 5 record, part A: the driver's command (full line), kernel stats of the headline workload, configs[1] and the 8-GPU shard as their own workloads
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
echo "== the driver's command"; timeout -k 10 900 python bench.py --gpus 1 > gpurun_out/r05_bench_driver_cmd.log 2>gpurun_out/r05_bench_driver_cmd.err || { tail -5 gpurun_out/r05_bench_driver_cmd.err; exit 1; }
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05_bench_driver_cmd.log").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "roofline", j["roofline"]["frac"], j["roofline"]["avg_launch_ms"], "verified", j.get("verified"))
for leg in ("c2", "shard8", "c5"):
    x = j.get(leg) or {}
    print(leg, x.get("queries_per_s"), x.get("ms_per_step"), (x.get("roofline") or {}).get("frac"), (x.get("roofline") or {}).get("avg_launch_ms"), x.get("steps"), x.get("error"), (x.get("exchange") or {}).get("merged_equals_direct"))
print("rerank", j["rerank_p50_ms"], (j.get("rerank") or {}).get("dp_shares"))
print("c4", {k: (j.get("c4") or {}).get(k) for k in ("table_rows", "error")}, (j.get("c4") or {}).get("p50_ms"))
PY
echo "== kernel stats, headline"; bash tools/gpu_prof_bench.sh r05_10m stats --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rerank --no-shard-legs || exit 1
echo "== kernel stats, configs[1] 1M x 768"; bash tools/gpu_prof_bench.sh r05_c2 stats --rows 1000000 --steps 200 --warmup 20 --no-cpu-baseline --no-rerank || exit 1
echo "== kernel stats, 8-GPU shard 1.25M x 768"; bash tools/gpu_prof_bench.sh r05_shard8 stats --rows 1250000 --steps 200 --warmup 20 --no-cpu-baseline --no-rerank || exit 1
