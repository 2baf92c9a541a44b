#!/bin/bash
# BASELINE configs[4] (10M x 1024 e4m3 rows, 1024 queries, k = 1000) and its 8-GPU shard (1.25M rows): the shipped kernel (fp16 MFMA on
# converted rows), and the TIMING-ONLY build whose matrix work uses the block-scaled fp8 MFMA with a hi + lo query (results invalid)
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_c5.log
for rows in 10000000 1250000; do
  for lib in "" libvf_f8t.so; do
    L=""; [ -n "$lib" ] && L="$PWD/veritasfi_amd/lib/$lib"
    nv=""; [ -n "$lib" ] && nv="--no-verify"
    steps=10; [ $rows -lt 2000000 ] && steps=40
    echo "== rows=$rows lib=${lib:-shipped}" >> gpurun_out/r03_c5.log
    VF_LIB_PATH=$L timeout -k 10 400 python3 bench.py --rows $rows --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --no-llm --no-c4 --steps $steps --warmup 3 $nv 2>/dev/null | grep -a "^{" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('q/s', d['value'], 'ms/step', d['ms_per_step'], 'kernel', r.get('kernel'), 'launch ms', r.get('avg_launch_ms'), 'achieved', r.get('achieved'), r.get('unit'), 'frac', r.get('frac'), 'cand/query', d['search_stats']['candidates_per_query'], 'reruns', d['search_stats']['exact_reruns_last_batch'])
" >> gpurun_out/r03_c5.log 2>&1 || echo failed >> gpurun_out/r03_c5.log
  done
done
cat gpurun_out/r03_c5.log
