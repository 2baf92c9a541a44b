#!/bin/bash
# round 5 record, part B: PMC traffic of configs[1], the 8-GPU shard and configs[4]; configs[4] kernel stats at both sizes
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
S="--steps 12 --warmup 3 --no-cpu-baseline --no-rerank --opt overlap_scans=0"
echo "== PMC 1M x 768"; bash tools/gpu_prof_bench.sh r05_c2_fetch FETCH_SIZE --rows 1000000 $S || exit 1
bash tools/gpu_prof_bench.sh r05_c2_write WRITE_SIZE --rows 1000000 $S || exit 1
echo "== PMC 1.25M x 768"; bash tools/gpu_prof_bench.sh r05_s8_fetch FETCH_SIZE --rows 1250000 $S || exit 1
bash tools/gpu_prof_bench.sh r05_s8_write WRITE_SIZE --rows 1250000 $S || exit 1
C5="--rows 10000000 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank"
echo "== configs[4]"; bash tools/gpu_prof_bench.sh r05_c5 stats $C5 --steps 24 --warmup 3 || exit 1
bash tools/gpu_prof_bench.sh r05_c5_fetch FETCH_SIZE $C5 --steps 6 --warmup 2 || exit 1
bash tools/gpu_prof_bench.sh r05_c5_write WRITE_SIZE $C5 --steps 6 --warmup 2 || exit 1
timeout -k 10 300 python bench.py --rows 1250000 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --steps 24 --warmup 3 > gpurun_out/r05_bench_c5_shard_1250k.log 2>/dev/null || exit 1
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05_bench_c5_shard_1250k.log").read().strip().splitlines()[-1]); r = j["roofline"]
print("c5 shard 1.25M: q/s", j["value"], "ms/step", j["ms_per_step"], "launch", r["avg_launch_ms"], "frac", r["frac"])
PY
python tools/pmc_traffic.py gpurun_out/r05_c2_fetch_FETCH_SIZE.csv gpurun_out/r05_c2_write_WRITE_SIZE.csv "k_scan2<2" 1000000 768 64 100 $(( (1000000 - 32768) * (768 * 2 + 4) )) gpurun_out/pmc_traffic_scan2_1000k.json "round 5 (tools/gpu_r05_record_b.sh)"
python tools/pmc_traffic.py gpurun_out/r05_s8_fetch_FETCH_SIZE.csv gpurun_out/r05_s8_write_WRITE_SIZE.csv "k_scan2<2" 1250000 768 64 100 $(( (1250000 - 32768) * (768 * 2 + 4) )) gpurun_out/pmc_traffic_scan2_1250k_r05.json "round 5 (tools/gpu_r05_record_b.sh)"
python tools/pmc_traffic.py gpurun_out/r05_c5_fetch_FETCH_SIZE.csv gpurun_out/r05_c5_write_WRITE_SIZE.csv "k_scan_wide8" 10000000 1024 1024 1000 $(( (10000000 - 32768) * (1024 + 4) )) gpurun_out/pmc_traffic_scan_wide8_c5_10Mx1024_r05.json "round 5 (tools/gpu_r05_record_b.sh)"
