#!/bin/bash
# small-shard (8-GPU shard size) bench + per-kernel breakdown.  Usage: gpurun -- bash tools/gpu_small_shard.sh [tag]
set -o pipefail
tag=${1:-x}
mkdir -p gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 300 python bench.py --rows 1250000 --steps 200 --warmup 20 --no-rerank --no-cpu-baseline > gpurun_out/bench_1250k_$tag.log 2>&1 || exit 1
tail -1 gpurun_out/bench_1250k_$tag.log
timeout -k 10 300 python bench.py --rows 1000000 --steps 200 --warmup 20 --no-rerank --no-cpu-baseline > gpurun_out/bench_1m_$tag.log 2>&1 || exit 1
tail -1 gpurun_out/bench_1m_$tag.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o p -- python bench.py --rows 1250000 --steps 200 --warmup 20 --no-rerank --no-cpu-baseline > gpurun_out/prof_$tag/run.log 2>&1 || exit 1
find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_1250k_$tag.csv
head -12 gpurun_out/kernel_stats_1250k_$tag.csv
find gpurun_out/prof_$tag -name "*.csv" ! -name "*stats*" -size +1M -delete
