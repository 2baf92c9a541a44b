#!/bin/bash
# A/B: non-temporal corpus loads (default build) vs plain loads (libvf_plain.so), stage depth, CU split
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/nt_sweep.log
run() {  # lib opts rows
  echo "== lib=$1 opts=[$2] rows=$3" >> gpurun_out/nt_sweep.log
  opts=""; for kv in $2; do opts="$opts --opt $kv"; done
  VF_LIB_PATH=$PWD/veritasfi_amd/lib/$1 VF_BENCH_DEPTH=2 timeout -k 10 200 python3 bench.py --gpus 1 --rows $3 --steps ${4:-300} --warmup 30 --no-cpu-baseline --no-rerank $opts 2>/dev/null \
    | python3 -c "import sys,json; [print({k: (d[k] if k!='roofline' else {kk: d[k][kk] for kk in ('frac','avg_launch_ms')}) for k in ('ms_per_step','roofline')}, d['search_stats']['candidates_per_query']) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" >> gpurun_out/nt_sweep.log 2>&1 || exit 1
}
for rep in 1 2; do
for rows in 1000000 1250000 10000000; do
  steps=300; [ $rows -gt 5000000 ] && steps=60
  run libvf_plain.so "aux_cus=0" $rows $steps
  run libveritasfi_hip.so "aux_cus=0" $rows $steps
  run libveritasfi_hip.so "aux_cus=0 scan_g=3" $rows $steps
  run libveritasfi_hip.so "aux_cus=32" $rows $steps
done; done
cat gpurun_out/nt_sweep.log
