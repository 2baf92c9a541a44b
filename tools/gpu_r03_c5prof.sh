#!/bin/bash
# rocprofv3 kernel stats of BASELINE configs[4] on one GPU (10M x 1024 e4m3, 1024 queries, k = 1000) and of its 8-GPU shard
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in "10000000 8 r03_kernel_stats_c5_10Mx1024" "1250000 30 r03_kernel_stats_c5_1250k"; do
  set -- $cfg
  rm -rf /tmp/prof_$3
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$3 -o out -- python3 $R/bench.py --rows $1 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --steps $2 --warmup 3 --no-cpu-baseline --no-rerank --no-llm --no-c4 --no-verify > $R/gpurun_out/$3.bench.log 2>/dev/null
  f=$(find /tmp/prof_$3 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -10 "$f" | cut -c1-220 > $R/gpurun_out/$3.csv
  python3 -c "import sys,json; d=[json.loads(l) for l in open('$R/gpurun_out/$3.bench.log') if l.startswith('{')][0]; r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_ms'], r['achieved'], r['frac'])"
  head -6 $R/gpurun_out/$3.csv | cut -c1-160
done
