#!/bin/bash
# k_scan2 with one asm statement per ring segment, against the previous commit's library: the retrieval tests, then configs[1] and a 640-wide corpus
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python3 -m pytest tests/test_gpu_retrieval.py -m gpu -x -q > gpurun_out/r06_ff_tests.log 2>&1 || { tail -30 gpurun_out/r06_ff_tests.log; exit 1; }
tail -2 gpurun_out/r06_ff_tests.log
L=gpurun_out/r06_dma_segment_asm_scan2_ab.log
: > $L
run() {  # label, lib, bench args
  local label="$1" lib="$2"; shift 2
  VF_LIB_PATH=$PWD/veritasfi_amd/lib/$lib timeout -k 10 300 python3 bench.py --gpus 1 --no-rerank --no-cpu-baseline --no-shard-legs --no-startup "$@" > gpurun_out/_ab.json 2>gpurun_out/_ab.err || { tail -5 gpurun_out/_ab.err; echo fail; exit 1; }
  python3 - "$label" <<'PY' >> $L
import json, sys
j = json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1]); r = j["roofline"]
print(f"{sys.argv[1]}: {j['ms_per_step']:.4f} ms/step  frac {r['frac']}  isolated {r.get('isolated_launch', {}).get('frac')}  kernel {r['kernel'][:24]}")
PY
}
for rep in 1 2 3; do
  for lib in libvf_prev.so libveritasfi_hip.so; do
    run "rep $rep [$lib] 1M x 768 fp16 (configs[1])" $lib --rows 1000000 --steps 200 --warmup 20
    run "rep $rep [$lib] 5M x 640 fp16" $lib --rows 5000000 --dim 640 --steps 60 --warmup 10
  done
done
cat $L
