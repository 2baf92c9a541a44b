#!/bin/bash
# the rows' LDS-DMA cache policy: nt (shipped) against sc1 nt / sc0 sc1 nt / sc0 nt / sc1, same box, alternating (variant libraries built by hand
# with -DVF_ROW_POLICY='"..."', not tracked)
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_row_policy_ab.log
: > $L
run() {  # label, lib, bench args
  local label="$1" lib="$2"; shift 2
  VF_LIB_PATH=$PWD/veritasfi_amd/lib/$lib timeout -k 10 300 python3 bench.py --gpus 1 --no-rerank --no-cpu-baseline --no-shard-legs --no-startup "$@" > gpurun_out/_ab.json 2>gpurun_out/_ab.err || { tail -5 gpurun_out/_ab.err; echo fail; exit 1; }
  python3 - "$label" <<'PY' >> $L
import json, sys
j = json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1]); r = j["roofline"]
print(f"{sys.argv[1]}: {j['ms_per_step']:.4f} ms/step  frac {r['frac']}  isolated {r.get('isolated_launch', {}).get('frac')}")
PY
}
for rep in 1 2; do
  for lib in libveritasfi_hip.so libvf_p_sc1_nt.so libvf_p_sc0_sc1_nt.so libvf_p_sc0_nt.so libvf_p_sc1.so; do
    run "rep $rep [$lib] 10M x 768 fp16" $lib --rows 10000000 --steps 40 --warmup 8
    run "rep $rep [$lib] 1.25M x 768 fp16" $lib --rows 1250000 --steps 200 --warmup 20
  done
done
cat $L
