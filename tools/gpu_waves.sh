#!/bin/bash
# sweep the scan grid (waves option) at shard sizes, exchange forced (1-rank RCCL), one JSON summary line each
mkdir -p gpurun_out
for w in 0 1984 1920 1792; do for r in 1250000 10000000; do
  echo "== waves=$w rows=$r"
  VF_BENCH_FORCE_EXCHANGE=1 timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --rows $r --steps 100 --warmup 20 --no-cpu-baseline --no-rerank --opt waves=$w > gpurun_out/_w.log 2>&1 || { tail -5 gpurun_out/_w.log; exit 1; }
  grep '^{' gpurun_out/_w.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['achieved'])"
done; done
