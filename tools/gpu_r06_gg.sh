#!/bin/bash
# candidate loop over the set pairs with a scalar register index (v_movrel), filter on the maximum of the scores: the retrieval tests, then A/B
# against the previous commit's library on one box, alternating
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python3 -m pytest tests/test_gpu_retrieval.py -m gpu -x -q > gpurun_out/r06_gg_tests.log 2>&1 || { tail -30 gpurun_out/r06_gg_tests.log; exit 1; }
tail -2 gpurun_out/r06_gg_tests.log
L=gpurun_out/r06_candidate_loop_movrel_ab.log
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
    run "rep $rep [$lib] 1.25M x 768 fp16" $lib --rows 1250000 --steps 200 --warmup 20
    run "rep $rep [$lib] 1M x 768 fp16 (configs[1])" $lib --rows 1000000 --steps 200 --warmup 20
    run "rep $rep [$lib] 10M x 768 fp16" $lib --rows 10000000 --steps 40 --warmup 8
    run "rep $rep [$lib] 10M x 768 e4m3" $lib --rows 10000000 --corpus-dtype fp8 --steps 40 --warmup 8
    run "rep $rep [$lib] 1.25M x 1024 e4m3" $lib --rows 1250000 --dim 1024 --corpus-dtype fp8 --steps 200 --warmup 20
  done
done
cat $L
