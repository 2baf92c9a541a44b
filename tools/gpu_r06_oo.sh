#!/bin/bash
# the new default kernels at full size, verified against the oracle over ALL rows in the same run: e4m3 10M x 1024 and 10M x 768, fp16 8M x 1024, 10M x 512
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_verified_full_size_new_defaults.log
: > $L
for spec in "10000000 1024 fp8" "10000000 768 fp8" "8000000 1024 f16" "10000000 512 f16"; do
  set -- $spec
  timeout -k 10 500 python3 bench.py --gpus 1 --rows $1 --dim $2 --corpus-dtype $3 --steps 20 --warmup 5 --no-rerank --no-shard-legs --no-startup --verify > gpurun_out/_v.json 2>gpurun_out/_v.err || { tail -5 gpurun_out/_v.err; echo fail; exit 1; }
  python3 - "$1 x $2 $3" <<'PY' >> $L
import json, sys
j = json.loads(open("gpurun_out/_v.json").read().strip().splitlines()[-1]); r = j["roofline"]; c = j.get("cpu_baseline") or {}
print(f"{sys.argv[1]}, batch 64, top-100: {j['value']:.0f} q/s  {j['ms_per_step']:.4f} ms/step  roofline {r['frac']}  kernel {r['kernel'][:18]}  verified {j.get('verified')}  (oracle: {c.get('value')} {c.get('unit')} on {c.get('cores')} cores, {c.get('sample')})")
PY
done
cat $L
