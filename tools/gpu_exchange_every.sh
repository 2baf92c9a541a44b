#!/bin/bash
# sweep --exchange-every at the 8-GPU shard size, exchange forced on one rank (1-rank RCCL under torchrun)
mkdir -p gpurun_out
for w in 1 2 4 1 2 4; do for r in 1250000; do
  echo "== exchange-every=$w rows=$r"
  VF_BENCH_FORCE_EXCHANGE=1 timeout -k 10 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --rows $r --steps 120 --warmup 24 --no-cpu-baseline --no-rerank --exchange-every $w > gpurun_out/_w.log 2>&1 || { tail -5 gpurun_out/_w.log; exit 1; }
  grep '^{' gpurun_out/_w.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['achieved'])"
done; done
