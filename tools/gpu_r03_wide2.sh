#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -x -p no:cacheprovider -k "wide or c5 or hostile" > gpurun_out/pytest_wide2.log 2>&1; rc=$?
tail -3 gpurun_out/pytest_wide2.log
if [ $rc -ne 0 ]; then grep -a "Error\|assert" gpurun_out/pytest_wide2.log | head -20; exit $rc; fi
: > gpurun_out/r03_c5b.log
for rows in 1250000 10000000; do
    steps=10; [ $rows -lt 2000000 ] && steps=40
    echo "== rows=$rows" >> gpurun_out/r03_c5b.log
    timeout -k 10 400 python3 bench.py --rows $rows --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --no-llm --no-c4 --steps $steps --warmup 3 2>/dev/null | grep -a "^{" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('q/s', d['value'], 'ms/step', d['ms_per_step'], 'launch ms', r.get('avg_launch_ms'), 'achieved', r.get('achieved'), r.get('unit'), 'frac', r.get('frac'), 'cand/query', d['search_stats']['candidates_per_query'], 'reruns', d['search_stats']['exact_reruns_last_batch'], 'verified', d.get('verified'))
" >> gpurun_out/r03_c5b.log 2>&1 || echo failed >> gpurun_out/r03_c5b.log
done
cat gpurun_out/r03_c5b.log
