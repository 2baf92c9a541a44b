#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/scan2_sweep5.log
run() {
  echo "== opts=[$1] rows=$2" >> gpurun_out/scan2_sweep5.log
  opts=""; for kv in $1; do opts="$opts --opt $kv"; done
  VF_BENCH_DEPTH=2 VF_BENCH_LAUNCH=1 VF_BENCH_FORCE_EXCHANGE=1 timeout -k 10 200 python3 bench.py --gpus 1 --rows $2 --steps 400 --warmup 30 --no-cpu-baseline --no-rerank --no-llm --no-c4 $opts 2>/dev/null \
    | python3 -c "import sys,json; [print(d['ms_per_step'], {kk: d['roofline'].get(kk) for kk in ('frac','avg_launch_ms')}, d['search_stats']['candidates_per_query'], d.get('verified')) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" >> gpurun_out/scan2_sweep5.log 2>&1 || echo failed >> gpurun_out/scan2_sweep5.log
}
for rows in 1250000; do
  run "" $rows
  run "sample_rows=24" $rows
  run "sample_rows=32" $rows
  run "sample_rows=48" $rows
  run "sample_rows=32 sample_grid=256" $rows
  run "" $rows
done
cat gpurun_out/scan2_sweep5.log
