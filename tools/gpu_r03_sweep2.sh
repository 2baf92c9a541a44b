#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/scan2_sweep2.log
run() {  # depth opts rows steps
  echo "== D=$1 opts=[$2] rows=$3" >> gpurun_out/scan2_sweep2.log
  opts=""; for kv in $2; do opts="$opts --opt $kv"; done
  VF_BENCH_DEPTH=$1 VF_BENCH_LAUNCH=1 VF_BENCH_FORCE_EXCHANGE=1 timeout -k 10 200 python3 bench.py --gpus 1 --rows $3 --steps $4 --warmup 30 --no-cpu-baseline --no-rerank $opts 2>/dev/null \
    | python3 -c "import sys,json; [print({k: (d[k] if k!='roofline' else {kk: d[k][kk] for kk in ('frac','avg_launch_ms')}) for k in ('ms_per_step','roofline')}, d['search_stats']['candidates_per_query'], d['search_stats']['exact_reruns_last_batch']) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" >> gpurun_out/scan2_sweep2.log 2>&1 || exit 1
}
for rows in 1250000 2500000 5000000; do
  steps=300; [ $rows -gt 2000000 ] && steps=120
  run 2 "aux_cus=0" $rows $steps
  run 2 "aux_cus=32 overlap_scans=1" $rows $steps
  run 3 "aux_cus=32 overlap_scans=1" $rows $steps
  run 4 "aux_cus=32 overlap_scans=1" $rows $steps
  run 3 "aux_cus=64 overlap_scans=1" $rows $steps
  run 3 "aux_cus=32 overlap_scans=1 sample_grid=32" $rows $steps
  run 3 "aux_cus=32 overlap_scans=1 sample_rows=8" $rows $steps
done
cat gpurun_out/scan2_sweep2.log
