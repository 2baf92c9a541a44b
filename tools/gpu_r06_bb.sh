#!/bin/bash
# k_scan2r's other fp16 shapes (1024 / 512 / 384): parity, then against the default kernel, whole chip + ordered scans and CU split + overlapping scans
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests/test_gpu_retrieval.py -m gpu -x -q -k "scan2r" > gpurun_out/r06_bb_tests.log 2>&1 || { tail -30 gpurun_out/r06_bb_tests.log; exit 1; }
tail -3 gpurun_out/r06_bb_tests.log
L=gpurun_out/r06_scan2r_fp16_other_widths_ab.log
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
  for spec in "1024 8000000 30 6" "1024 1250000 200 20" "512 10000000 30 6" "384 10000000 30 6" "512 1250000 200 20"; do
    set -- $spec
    st="--steps $3 --warmup $4"
    run "rep $rep $2 x $1 fp16 [default]" --rows $2 --dim $1 $st
    run "rep $rep $2 x $1 fp16 [k_scan2r, whole chip, ordered]" --rows $2 --dim $1 $st --opt scan_impl=5 --opt aux_cus=0 --opt overlap_scans=0
    run "rep $rep $2 x $1 fp16 [k_scan2r + its sample pass, split + overlap]" --rows $2 --dim $1 $st --opt scan_impl=5 --opt sample_impl=1 --opt aux_cus=32 --opt overlap_scans=1
  done
done
cat $L
