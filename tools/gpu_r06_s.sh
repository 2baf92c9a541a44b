#!/bin/bash
# what bounds k_scan2r on e4m3 rows: the conversions or the matrix instructions?  (test variant of the library, INVALID results, timing only)
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_fp8_scan2r_what_bounds.log
: > $L
export VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_test.so
for dim in 768 1024; do
  for o in "--opt scan_impl=5" "--opt scan_impl=5 --opt debug=64" "--opt scan_impl=5 --opt debug=32" "--opt scan_impl=5 --opt debug=96"; do
    timeout -k 10 300 python3 bench.py --gpus 1 --rows 10000000 --dim $dim --corpus-dtype fp8 --steps 40 --warmup 8 --no-rerank --no-cpu-baseline --no-shard-legs --no-startup $o > gpurun_out/_ab.json 2>gpurun_out/_ab.err || { tail -5 gpurun_out/_ab.err; echo fail; exit 1; }
    python3 - $dim "$o" <<'PY' >> $L
import json, sys
j = json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1]); r = j["roofline"]
print(f"10M x {sys.argv[1]} e4m3, batch 64 [{sys.argv[2]}]: {j['ms_per_step']:.4f} ms/step  frac {r['frac']}  kernel {r['kernel'][:24]}")
PY
  done
done
cat $L
