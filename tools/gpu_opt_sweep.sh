#!/bin/bash
# usage: gpu_opt_sweep.sh <optname> "<values>" "<rows list>"   (no exchange; one line per run)
mkdir -p gpurun_out
for v in $2; do for r in $3; do
  timeout -k 10 150 python bench.py --rows $r --steps 150 --warmup 20 --no-cpu-baseline --no-rerank $EXTRA --opt $1=$v > gpurun_out/_o.log 2>&1 || { tail -5 gpurun_out/_o.log; exit 1; }
  grep '^{' gpurun_out/_o.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1=$v rows=$r', d['ms_per_step'], d['value'], d['roofline']['achieved'], d['config'].get('candidates_per_query'))"
done; done
