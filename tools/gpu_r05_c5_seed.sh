#!/bin/bash
# round 5: configs[4] -- how the candidate volume (10.2 k per query for k = 1000) and the launch respond to a larger sample and a finer refresh
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
L=$R/gpurun_out/r05_c5_seed_and_refresh.log
: > $L
C5="--rows 10000000 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --steps 12 --warmup 3"
for opt in "" "--opt sample_rows=32" "--opt sample_rows=64" "--opt refresh_every=64" "--opt refresh_every=32" "--opt sample_rows=64 --opt refresh_every=64" "--opt refresh_every=256" ""; do
  echo "== $opt" >> $L
  timeout -k 10 200 python bench.py $C5 $opt 2>/dev/null | tail -1 > /tmp/c5.json || exit 1
  python - >> $L <<'PY'
import json
j = json.loads(open("/tmp/c5.json").read()); r = j["roofline"]
print("   q/s", j["value"], "ms/step", j["ms_per_step"], "launch", r["avg_launch_ms"], "frac", r["frac"], "cand/query", (j.get("search_stats") or {}).get("candidates_per_query"), "reruns", (j.get("search_stats") or {}).get("exact_reruns_last_batch"))
PY
done
cat $L
