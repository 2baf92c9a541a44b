#!/bin/bash
# after the non-temporal row loads: do the dispatch rules still hold?  (CU split vs whole chip, k_scan2 vs k_scan2r at 1M, e4m3 rows)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python3 -m pytest tests/test_gpu_retrieval.py -m gpu -x -q > gpurun_out/r06_kk_tests.log 2>&1 || { tail -30 gpurun_out/r06_kk_tests.log; exit 1; }
tail -2 gpurun_out/r06_kk_tests.log
L=gpurun_out/r06_rules_after_nt.log
: > $L
run() {  # label, bench args
  local label="$1"; shift
  timeout -k 10 300 python3 bench.py --gpus 1 --no-rerank --no-cpu-baseline --no-shard-legs --no-startup "$@" > gpurun_out/_ab.json 2>gpurun_out/_ab.err || { tail -5 gpurun_out/_ab.err; echo fail; exit 1; }
  python3 - "$label" <<'PY' >> $L
import json, sys
j = json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1]); r = j["roofline"]
print(f"{sys.argv[1]}: {j['ms_per_step']:.4f} ms/step  frac {r['frac']}  isolated {r.get('isolated_launch', {}).get('frac')}  kernel {r['kernel'][:24]}")
PY
}
for rep in 1 2; do
  run "rep $rep 10M x 768 fp16 [default: split 32 + overlap]" --rows 10000000 --steps 40 --warmup 8
  run "rep $rep 10M x 768 fp16 [whole chip, ordered, k_scan2r]" --rows 10000000 --steps 40 --warmup 8 --opt aux_cus=0 --opt overlap_scans=0 --opt scan_impl=5
  run "rep $rep 10M x 768 fp16 [whole chip, ordered, k_scan2]" --rows 10000000 --steps 40 --warmup 8 --opt aux_cus=0 --opt overlap_scans=0 --opt scan_impl=4
  run "rep $rep 1M x 768 fp16 [k_scan2: default]" --rows 1000000 --steps 200 --warmup 20
  run "rep $rep 1M x 768 fp16 [k_scan2r: scan_impl=5]" --rows 1000000 --steps 200 --warmup 20 --opt scan_impl=5
  run "rep $rep 10M x 768 e4m3 [default: whole chip, ordered]" --rows 10000000 --corpus-dtype fp8 --steps 40 --warmup 8
  run "rep $rep 10M x 768 e4m3 [split 32 + overlap]" --rows 10000000 --corpus-dtype fp8 --steps 40 --warmup 8 --opt aux_cus=32 --opt overlap_scans=1
  run "rep $rep 10M x 768 e4m3 [k_scan]" --rows 10000000 --corpus-dtype fp8 --steps 40 --warmup 8 --opt scan_impl=1
  run "rep $rep 8M x 1024 fp16 [default]" --rows 8000000 --dim 1024 --steps 40 --warmup 8
  run "rep $rep 8M x 1024 fp16 [k_scan]" --rows 8000000 --dim 1024 --steps 40 --warmup 8 --opt scan_impl=1
done
cat $L
