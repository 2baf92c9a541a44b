#!/bin/bash
# small-shard (8-GPU shard size) bench + per-kernel breakdown.  Usage: gpurun -- bash tools/gpu_small_shard.sh [tag] [extra bench args]
set -o pipefail
tag=${1:-x}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/prof_$tag
timeout -k 10 300 python3 bench.py --rows 1250000 --steps 200 --warmup 20 --no-rerank --no-cpu-baseline "$@" > gpurun_out/bench_1250k_$tag.log 2>&1 || exit 1
tail -1 gpurun_out/bench_1250k_$tag.log | cut -c1-420
timeout -k 10 300 python3 bench.py --rows 1000000 --steps 200 --warmup 20 --no-rerank --no-cpu-baseline "$@" > gpurun_out/bench_1m_$tag.log 2>&1 || exit 1
tail -1 gpurun_out/bench_1m_$tag.log | cut -c1-420
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$tag -o p -- python3 $REPO/bench.py --rows 1250000 --steps 200 --warmup 20 --no-rerank --no-cpu-baseline "$@" > $REPO/gpurun_out/prof_$tag/run.log 2>&1 || exit 1
cd $REPO
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/kernel_stats_1250k_$tag.csv
head -12 $f | cut -c1-200
find gpurun_out/prof_$tag -name "*.csv" ! -name "*stats*" -delete
