#!/bin/bash
# round 5: the 4-wave form of k_scan_wide8 with its siblings paced (wide_sync) -- does lock-step bring the two-workgroups-per-CU form home?
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r05_w8sync.log
: > $L
for opts in "wide8_waves=4" "wide8_waves=4 wide_sync=0" "wide8_waves=4 wide_sync=2" "wide8_waves=8 wide_sync=1"; do
  o=""; for x in $opts; do o="$o --opt $x"; done
  echo "== rows 10000000 $opts" | tee -a $L
  timeout -k 10 300 python bench.py --rows 10000000 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --steps 6 --warmup 2 $o >> $L 2>gpurun_out/r05_w8sync.err || { tail -20 gpurun_out/r05_w8sync.err; exit 1; }
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_w8sync.log"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        j = json.loads(l); r = j["roofline"]
        print("  ms/step", j["ms_per_step"], "launch", r["avg_launch_ms"], "TF", r["achieved"], "frac", r["frac"])
PY
