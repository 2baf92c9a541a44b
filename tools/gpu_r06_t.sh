#!/bin/bash
# k_scan2r with the B fragments pinned to accumulator registers and a segment's LDS reads issued together: parity, then A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests/test_gpu_retrieval.py -m gpu -x -q -k "scan2r or scan_kernels_agree_on_fp8" > gpurun_out/r06_scan2r_areg_tests.log 2>&1 || { tail -30 gpurun_out/r06_scan2r_areg_tests.log; exit 1; }
tail -3 gpurun_out/r06_scan2r_areg_tests.log
L=gpurun_out/r06_scan2r_areg_ab.log
: > $L
export VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_test.so
run() {  # label, bench args
  local label="$1"; shift
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 40 --warmup 8 --no-rerank --no-cpu-baseline --no-shard-legs --no-startup "$@" > gpurun_out/_ab.json 2>gpurun_out/_ab.err || { tail -5 gpurun_out/_ab.err; echo fail; exit 1; }
  python3 - "$label" <<'PY' >> $L
import json, sys
j = json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1]); r = j["roofline"]
print(f"{sys.argv[1]}: {j['ms_per_step']:.4f} ms/step  frac {r['frac']}  isolated {r.get('isolated_launch', {}).get('frac')}  kernel {r['kernel'][:24]}")
PY
}
for rep in 1 2; do
  run "rep $rep 10M x 768 fp16 [first form: debug=1024]" --rows 10000000 --opt debug=1024
  run "rep $rep 10M x 768 fp16 [pinned]" --rows 10000000
  run "rep $rep 1.25M x 768 fp16 [first form]" --rows 1250000 --steps 200 --warmup 20 --opt debug=1024
  run "rep $rep 1.25M x 768 fp16 [pinned]" --rows 1250000 --steps 200 --warmup 20
  run "rep $rep 10M x 768 e4m3 [k_scan]" --rows 10000000 --corpus-dtype fp8
  run "rep $rep 10M x 768 e4m3 [k_scan2r pinned]" --rows 10000000 --corpus-dtype fp8 --opt scan_impl=5
  run "rep $rep 10M x 768 e4m3 [k_scan2r pinned, split + overlap]" --rows 10000000 --corpus-dtype fp8 --opt scan_impl=5 --opt aux_cus=32 --opt overlap_scans=1 --opt sample_impl=1
  run "rep $rep 10M x 1024 e4m3 [k_scan]" --rows 10000000 --dim 1024 --corpus-dtype fp8
  run "rep $rep 10M x 1024 e4m3 [k_scan2r pinned]" --rows 10000000 --dim 1024 --corpus-dtype fp8 --opt scan_impl=5
  run "rep $rep 10M x 1024 e4m3 [k_scan2r pinned, split + overlap]" --rows 10000000 --dim 1024 --corpus-dtype fp8 --opt scan_impl=5 --opt aux_cus=32 --opt overlap_scans=1 --opt sample_impl=1
done
cat $L
