#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_retrieval.py -k "scan2r or scan_kernels_agree_and_match or fuzz_across or small_path or fused_path_bit_exact or few_queries or deep_searches" -x -q -m gpu > gpurun_out/r06_n_tests.log 2>&1 || { tail -40 gpurun_out/r06_n_tests.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r06_n_tests.log
L=gpurun_out/r06_sample2r_ab.log
: > $L
for rep in 1 2 3; do
  for impl in 0 1; do
    for rows in 1000000 1250000 10000000; do
      steps=300; [ $rows = 10000000 ] && steps=60
      timeout -k 10 200 python3 bench.py --gpus 1 --rows $rows --steps $steps --warmup 20 --no-rerank --no-cpu-baseline --no-shard-legs --no-startup --opt sample_impl=$impl > gpurun_out/_ab.json 2>/dev/null || { echo fail; exit 1; }
      python3 - $rep $impl $rows <<'PY' >> $L
import json, sys
j = json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1]); r = j["roofline"]
print(f"rep {sys.argv[1]} sample_impl {sys.argv[2]} rows {sys.argv[3]:>8s}: {j['ms_per_step']:.4f} ms/step  p50 {j['p50_ms_per_step']}  interval frac {r['frac']}  cand/q {j['search_stats']['candidates_per_query']}")
PY
    done
  done
done
cat $L
