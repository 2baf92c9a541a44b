#!/bin/bash
# fp16 rows of other widths at 64 queries: which kernel serves them and at what fraction (before giving k_scan2r more shapes)
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_fp16_other_widths.log
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
for spec in "384 10000000" "512 10000000" "1024 8000000" "1536 5000000"; do
  set -- $spec
  run "$2 x $1 fp16 [default]" --rows $2 --dim $1 --steps 30 --warmup 6
  run "$2 x $1 fp16 [k_scan: scan_impl=1]" --rows $2 --dim $1 --steps 30 --warmup 6 --opt scan_impl=1
done
cat $L
