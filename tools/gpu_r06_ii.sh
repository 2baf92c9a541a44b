#!/bin/bash
# corpus rows by LDS-DMA with the non-temporal policy (read once per batch) against the default policy: same box, alternating
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_dma_nt_ab.log
: > $L
run() {  # label, lib, bench args
  local label="$1" lib="$2"; shift 2
  VF_LIB_PATH=$PWD/veritasfi_amd/lib/$lib timeout -k 10 300 python3 bench.py --gpus 1 --no-rerank --no-cpu-baseline --no-shard-legs --no-startup "$@" > gpurun_out/_ab.json 2>gpurun_out/_ab.err || { tail -5 gpurun_out/_ab.err; echo fail; exit 1; }
  python3 - "$label" <<'PY' >> $L
import json, sys
j = json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1]); r = j["roofline"]
print(f"{sys.argv[1]}: {j['ms_per_step']:.4f} ms/step  frac {r['frac']}  isolated {r.get('isolated_launch', {}).get('frac')}  verified {j.get('verified')}  kernel {r['kernel'][:24]}")
PY
}
for rep in 1 2 3; do
  for lib in libvf_prev.so libveritasfi_hip.so; do
    run "rep $rep [$lib] 10M x 768 fp16" $lib --rows 10000000 --steps 40 --warmup 8 --verify
    run "rep $rep [$lib] 1.25M x 768 fp16" $lib --rows 1250000 --steps 200 --warmup 20
    run "rep $rep [$lib] 1M x 768 fp16 (configs[1])" $lib --rows 1000000 --steps 200 --warmup 20
    run "rep $rep [$lib] 10M x 1024 e4m3" $lib --rows 10000000 --dim 1024 --corpus-dtype fp8 --steps 40 --warmup 8
  done
done
cat $L
