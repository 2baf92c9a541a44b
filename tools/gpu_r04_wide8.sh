#!/bin/bash
# round 4: configs[4] (10M x 1024 e4m3, 1024 queries, k = 1000) and its 8-GPU shard with the fp16 instruction (k_scan_wide) and the fp8 one (k_scan_wide8)
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_wide8.log
: > $L
timeout -k 10 400 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -p no:cacheprovider -x -k "fp8_matrix_instruction or wide_scan" >> $L 2>&1; rc=$?
tail -3 $L
[ $rc -ne 0 ] && tail -40 $L && exit $rc
for rows in 1250000 10000000; do VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_stamps.so timeout -k 10 300 python tools/stamps_wide8.py $rows 2>&1 | grep -v amdgpu.ids | tee -a $L; done
for rows in 1250000 10000000; do
  C5="--rows $rows --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --steps 10 --warmup 2"
  for m in 0 1; do
    echo "== rows $rows wide_mfma=$m" | tee -a $L
    timeout -k 10 400 python bench.py $C5 --opt wide_mfma=$m >> $L 2>gpurun_out/r04_wide8.err || { tail -20 gpurun_out/r04_wide8.err; exit 1; }
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r04_wide8.log"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        j = json.loads(l)
        r = j["roofline"]
        print("  value", j["value"], "ms/step", j["ms_per_step"], "launch", r["avg_launch_ms"], "TF", r["achieved"], "frac", r["frac"], r["kernel"][:24], j.get("search_stats"))
PY
