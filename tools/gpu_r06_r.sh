#!/bin/bash
# k_scan2r on e4m3 rows (768 / 1024 elements, batch 64): parity first, then 10M rows against k_scan (the default for e4m3 rows)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests/test_gpu_retrieval.py -m gpu -x -q -k "scan2r or scan_kernels_agree_on_fp8" > gpurun_out/r06_fp8_scan2r_tests.log 2>&1 || { tail -30 gpurun_out/r06_fp8_scan2r_tests.log; exit 1; }
tail -3 gpurun_out/r06_fp8_scan2r_tests.log
L=gpurun_out/r06_fp8_scan2r_ab.log
: > $L
for rep in 1 2; do
  for dim in 768 1024; do
    for o in "" "--opt scan_impl=5" "--opt scan_impl=5 --opt sample_impl=1" "--opt scan_impl=5 --opt aux_cus=32 --opt overlap_scans=1" "--opt scan_impl=5 --opt sample_impl=1 --opt aux_cus=32 --opt overlap_scans=1"; do
      timeout -k 10 300 python3 bench.py --gpus 1 --rows 10000000 --dim $dim --corpus-dtype fp8 --steps 40 --warmup 8 --no-rerank --no-cpu-baseline --no-shard-legs --no-startup $o > gpurun_out/_ab.json 2>/dev/null || { echo fail; exit 1; }
      python3 - $rep $dim "$o" <<'PY' >> $L
import json, sys
j = json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1]); r = j["roofline"]
print(f"rep {sys.argv[1]} 10M x {sys.argv[2]} e4m3, batch 64 [{sys.argv[3] or 'default'}]: {j['ms_per_step']:.4f} ms/step  frac {r['frac']}  isolated {r.get('isolated_launch', {}).get('frac')}  kernel {r['kernel'][:24]}")
PY
    done
  done
done
cat $L
