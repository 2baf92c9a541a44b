#!/bin/bash
# kernel trace of the 8-GPU shard step (1.25M x 768, batch 64) -> timeline of a few steps
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/trace_small
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/trace_small -o t -- python3 $REPO/bench.py --rows 1250000 --steps 100 --warmup 20 --no-rerank --no-cpu-baseline "$@" > $REPO/gpurun_out/trace_small/run.log 2>&1 || { tail -5 $REPO/gpurun_out/trace_small/run.log; exit 1; }
cd $REPO
f=$(find gpurun_out/trace_small -name "*kernel_trace.csv" | head -1)
python3 tools/trace_span.py $f k_scan2 100
n=$(wc -l < $f)
python3 tools/trace_timeline.py $f $((n - 800)) 44 | tee gpurun_out/r04_small_timeline.log
tail -1 gpurun_out/trace_small/run.log | cut -c1-300
rm -f $f
