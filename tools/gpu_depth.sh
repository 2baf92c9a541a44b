#!/bin/bash
# pipeline depth sweep: ms/step, scan ms (HIP events), ratio
mkdir -p gpurun_out
for rep in 1 2; do for dpt in 2 3 4; do for r in 1250000 10000000; do
  VF_BENCH_DEPTH=$dpt timeout -k 10 150 python bench.py --rows $r --steps 150 --warmup 20 --no-cpu-baseline --no-rerank > gpurun_out/_o.log 2>&1 || { tail -5 gpurun_out/_o.log; exit 1; }
  grep '^{' gpurun_out/_o.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; sm=r.get('kernel_ms') or 0; print('depth=$dpt rows=$r', d['ms_per_step'], d['value'], r['achieved'], r)"
done; done; done
