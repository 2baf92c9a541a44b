#!/bin/bash
# configs[4] (k_scan_wide8): corpus rows by LDS-DMA with the non-temporal policy against the default, same box, alternating
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_dma_nt_wide8_ab.log
: > $L
for rep in 1 2; do
  for lib in libvf_prev.so libveritasfi_hip.so; do
    VF_LIB_PATH=$PWD/veritasfi_amd/lib/$lib timeout -k 10 400 python3 bench.py --gpus 1 --rows 10000000 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --no-startup --no-shard-legs --steps 24 --warmup 3 > gpurun_out/_ab.json 2>gpurun_out/_ab.err || { tail -5 gpurun_out/_ab.err; echo fail; exit 1; }
    python3 - "rep $rep [$lib] configs[4] 10M x 1024 e4m3, 1024 queries, k = 1000" <<'PY' >> $L
import json, sys
j = json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1]); r = j["roofline"]
print(f"{sys.argv[1]}: {j['ms_per_step']:.4f} ms/step  {j['value']:.0f} q/s  frac {r['frac']}  kernel {r['kernel'][:24]}")
PY
  done
done
cat $L
