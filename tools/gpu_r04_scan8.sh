#!/bin/bash
# e4m3-resident corpus at batch 64: k_scan (default) against k_scan2 with converted rows (scan_impl 3)
# (scan_impl 4, k_scan2 on the fp8 instruction, was measured with this script and removed later in round 4)
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_scan8.log
: > $L
timeout -k 10 300 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -p no:cacheprovider -x -k "agree_on_fp8_rows or e4m3 or fp8" >> $L 2>&1; rc=$?
tail -3 $L
[ $rc -ne 0 ] && tail -40 $L && exit $rc
for shape in "10000000 768" "10000000 1024" "1250000 1024"; do
  set -- $shape
  for impl in 1 3; do
    echo "== rows $1 dim $2 scan_impl=$impl" | tee -a $L
    timeout -k 10 300 python bench.py --rows $1 --dim $2 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --steps 40 --warmup 5 --opt scan_impl=$impl >> $L 2>gpurun_out/r04_scan8.err || { tail -20 gpurun_out/r04_scan8.err; exit 1; }
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r04_scan8.log"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        j = json.loads(l); r = j["roofline"]
        print("  value", j["value"], "ms/step", j["ms_per_step"], "launch", r.get("avg_launch_ms"), "GB/s", r.get("achieved"), "frac", r.get("frac"), r.get("kernel", "")[:28], j.get("search_stats"))
PY
