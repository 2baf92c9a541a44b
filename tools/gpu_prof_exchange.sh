#!/bin/bash
# rocprofv3 kernel stats of the bench at the 8-GPU shard size with the exchange forced (single process, no torchrun:
# bench.py initialises a 1-rank process group itself when RANK is set)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/exch
export VF_BENCH_FORCE_EXCHANGE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29544
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/exch -o exch -- python3 $REPO/bench.py --rows 1250000 --steps 100 --warmup 10 --no-cpu-baseline --no-rerank --exchange-every ${1:-1} > $REPO/gpurun_out/exch/run.log 2>&1
tail -1 $REPO/gpurun_out/exch/run.log | cut -c1-200
