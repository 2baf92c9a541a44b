#!/bin/bash
# kernel mix of the configured LLM re-ranker's forward (gemma-2b shape, 100 ragged inputs of 460..1052 tokens, left-padded, packed)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_llm
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_llm -o out -- python3 $R/tools/bench_decoder.py --shape gemma-2b --batch 100 --tokens 1056 --ragged 460 --iters 3 > $R/gpurun_out/llm_prof.log 2>&1
tail -2 $R/gpurun_out/llm_prof.log
f=$(find /tmp/prof_llm -name "*kernel_stats.csv" | head -1)
head -16 "$f" | cut -c1-230 | tee $R/gpurun_out/r03_kernel_stats_llm_gemma2b_100.csv
