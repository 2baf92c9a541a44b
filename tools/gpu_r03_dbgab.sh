#!/bin/bash
# where does the candidate path's time go?  bench timing with the scan's debug switches (results are wrong with them: timing only)
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/dbg_ab.log
run() {  # lib opts rows steps
  echo "== lib=$1 opts=[$2] rows=$3" >> gpurun_out/dbg_ab.log
  opts=""; for kv in $2; do opts="$opts --opt $kv"; done
  lib=""; [ -n "$1" ] && lib="$PWD/veritasfi_amd/lib/$1"
  VF_LIB_PATH=$lib VF_BENCH_DEPTH=2 VF_BENCH_LAUNCH=1 VF_BENCH_FORCE_EXCHANGE=1 timeout -k 10 200 python3 bench.py --gpus 1 --rows $3 --steps $4 --warmup 30 --no-cpu-baseline --no-rerank --no-verify --no-llm --no-c4 $opts 2>/dev/null \
    | python3 -c "import sys,json; [print(d['ms_per_step'], {kk: d['roofline'].get(kk) for kk in ('frac','avg_launch_ms')}, (d['roofline'].get('isolated_launch') or {}).get('avg_launch_ms'), d['search_stats']['candidates_per_query']) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" >> gpurun_out/dbg_ab.log 2>&1 || echo failed >> gpurun_out/dbg_ab.log
}
for rows in 1250000; do
  run "" "" $rows 400
  run "" "debug=4" $rows 400
  run "" "debug=8" $rows 400
  run "" "debug=2" $rows 400
  run libvf_nosvc.so "" $rows 400
  run libvf_nosvc.so "debug=4" $rows 400
  run libvf_nosvc.so "debug=8" $rows 400
  run libvf_nosvc.so "debug=2" $rows 400
  run "" "aux_cus=0 overlap_scans=0" $rows 400
  run "" "aux_cus=0 overlap_scans=0 debug=4" $rows 400
done
cat gpurun_out/dbg_ab.log
