#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_c5c.log
for re in 8 32 128 256; do
    echo -n "refresh_every=$re: " >> gpurun_out/r03_c5c.log
    timeout -k 10 400 python3 bench.py --rows 1250000 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --no-llm --no-c4 --no-verify --steps 40 --warmup 3 --opt refresh_every=$re 2>/dev/null | grep -a "^{" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('ms/step', d['ms_per_step'], 'launch ms', r.get('avg_launch_ms'), 'frac', r.get('frac'), 'cand/query', d['search_stats']['candidates_per_query'], 'reruns', d['search_stats']['exact_reruns_last_batch'])
" >> gpurun_out/r03_c5c.log 2>&1 || echo failed >> gpurun_out/r03_c5c.log
done
cat gpurun_out/r03_c5c.log
