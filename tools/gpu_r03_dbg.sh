#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/dbg_sweep.log
run() {
  echo "== opts=[$1] rows=$2" >> gpurun_out/dbg_sweep.log
  opts=""; for kv in $1; do opts="$opts --opt $kv"; done
  VF_BENCH_DEPTH=2 timeout -k 10 200 python3 bench.py --gpus 1 --rows $2 --steps 300 --warmup 30 --no-cpu-baseline --no-rerank $opts 2>/dev/null \
    | python3 -c "import sys,json; [print(d['ms_per_step'], {kk: d['roofline'].get(kk) for kk in ('frac','avg_launch_ms')}, (d['roofline'].get('isolated_launch') or {}).get('avg_launch_ms'), d['search_stats']['candidates_per_query']) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" >> gpurun_out/dbg_sweep.log 2>&1 || echo failed >> gpurun_out/dbg_sweep.log
}
for rows in 1250000; do
  run "aux_cus=32 overlap_scans=0" $rows
  run "aux_cus=32 overlap_scans=0 debug=4" $rows
  run "aux_cus=0 overlap_scans=0" $rows
  run "aux_cus=0 overlap_scans=0 debug=4" $rows
  run "aux_cus=32 overlap_scans=0 sample_rows=32" $rows
  run "aux_cus=32 overlap_scans=1 sample_rows=32" $rows
  run "aux_cus=32 overlap_scans=1 refresh_every=32" $rows
  run "aux_cus=32 overlap_scans=1" $rows
done
cat gpurun_out/dbg_sweep.log
