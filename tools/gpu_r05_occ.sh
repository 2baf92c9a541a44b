#!/bin/bash
# round 5: (1) the new host-side GPU tests; (2) resident workgroups per CU of k_scan_wide8<4> against its LDS size; (3) the 4-wave form with smaller candidate stages
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r05_occ.log
: > $L
timeout -k 10 500 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -p no:cacheprovider -x -k "reuse_their_workspace or rows_by_id or single_process_sharded or g6 or drop_in" >> $L 2>&1; rc=$?
tail -3 $L
[ $rc -ne 0 ] && tail -60 $L && exit $rc
python - <<'PY' 2>&1 | tee -a $L
import ctypes
from veritasfi_amd import _ffi
L = _ffi.lib()
for waves, caps in ((8, (3584,)), (4, (896, 768, 640, 512, 256, 64))):
    for cap in caps:
        print(f"k_scan_wide8<{waves}> stage_cap {cap}: resident workgroups per CU = {L.vf_debug_wide8_occupancy(waves, cap)}")
PY
for stage in 896 512 128; do
  C5="--rows 10000000 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --steps 6 --warmup 2 --opt wide8_waves=4 --opt wide8_stage=$stage"
  echo "== rows 10000000 wide8_waves=4 wide8_stage=$stage" | tee -a $L
  timeout -k 10 300 python bench.py $C5 >> $L 2>gpurun_out/r05_occ.err || { tail -20 gpurun_out/r05_occ.err; exit 1; }
done
python - <<'PY'
import json
for l in open("gpurun_out/r05_occ.log"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        j = json.loads(l); r = j["roofline"]
        print("  value", j["value"], "ms/step", j["ms_per_step"], "launch", r["avg_launch_ms"], "TF", r["achieved"], "frac", r["frac"], j.get("search_stats"))
PY
