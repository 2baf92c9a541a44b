#!/bin/bash
# few queries, deep (the reference's own call shape: 1-4 queries, k = 1000 / 2048): k_scan2r<1> against the kernels it replaced
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_scan2r_few_queries.log
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
  for spec in "768 10000000 f16 30 6" "768 1250000 f16 200 20" "1024 8000000 f16 30 6" "1024 10000000 fp8 30 6"; do
    set -- $spec
    st="--steps $4 --warmup $5 --batch 4 --k 1000 --corpus-dtype $3"
    run "rep $rep $2 x $1 $3, 4 queries, k = 1000 [default]" --rows $2 --dim $1 $st
    run "rep $rep $2 x $1 $3, 4 queries, k = 1000 [never k_scan2r: scan_impl=4]" --rows $2 --dim $1 $st --opt scan_impl=4
  done
done
cat $L
