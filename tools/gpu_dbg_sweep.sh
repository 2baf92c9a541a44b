#!/bin/bash
# usage: gpu_dbg_sweep.sh "<debug values>" "<rows list>"  -- bench lines per debug option value
mkdir -p gpurun_out
for v in $1; do for r in $2; do
  timeout -k 10 150 python bench.py --rows $r --steps 150 --warmup 20 --no-cpu-baseline --no-rerank --opt debug=$v > gpurun_out/_o.log 2>&1 || { tail -5 gpurun_out/_o.log; exit 1; }
  grep '^{' gpurun_out/_o.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('debug=$v rows=$r', d['ms_per_step'], d['value'], d['roofline']['achieved'])"
done; done
