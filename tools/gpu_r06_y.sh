#!/bin/bash
# after the filter rewrite: which kernel for e4m3 rows (k_scan vs k_scan2r, whole chip vs CU split + overlapping scans), and k_scan2 vs k_scan2r at 1M fp16 rows
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_after_filter_kernel_choice.log
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
  for dim in 768 1024; do
    for rows in 1000000 1250000 10000000; do
      st="--steps 200 --warmup 20"; [ $rows = 10000000 ] && st="--steps 40 --warmup 8"
      run "rep $rep ${rows} x $dim e4m3 [k_scan]" --rows $rows --dim $dim --corpus-dtype fp8 $st
      run "rep $rep ${rows} x $dim e4m3 [k_scan2r]" --rows $rows --dim $dim --corpus-dtype fp8 $st --opt scan_impl=5
      run "rep $rep ${rows} x $dim e4m3 [k_scan2r + its sample pass]" --rows $rows --dim $dim --corpus-dtype fp8 $st --opt scan_impl=5 --opt sample_impl=1
      run "rep $rep ${rows} x $dim e4m3 [k_scan2r + its sample pass, split + overlap]" --rows $rows --dim $dim --corpus-dtype fp8 $st --opt scan_impl=5 --opt sample_impl=1 --opt aux_cus=32 --opt overlap_scans=1
    done
  done
  run "rep $rep 1M x 768 fp16 [k_scan2: scan_impl=4]" --rows 1000000 --steps 200 --warmup 20 --opt scan_impl=4
  run "rep $rep 1M x 768 fp16 [k_scan2r: scan_impl=5]" --rows 1000000 --steps 200 --warmup 20 --opt scan_impl=5
done
cat $L
