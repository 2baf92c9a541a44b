#!/bin/bash
# CU-partitioned pipeline: A/B of the split at the 8-GPU shard size with the exchange forced
set -o pipefail
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O2 -o /tmp/cu_mask_probe tools/ubench/cu_mask_probe.hip 2>/dev/null && timeout -k 5 60 /tmp/cu_mask_probe > gpurun_out/cu_mask_probe.log 2>&1
cat gpurun_out/cu_mask_probe.log
: > gpurun_out/split_sweep.log
for cfg in ${VF_SWEEP:-"aux_cus=0,D=2" "aux_cus=32,D=2" "aux_cus=32,D=3" "aux_cus=32,D=3,overlap_scans=1" "aux_cus=64,D=3" "aux_cus=32,D=3,sample_grid=64" "aux_cus=32,D=3,sample_grid=256" "aux_cus=0,D=2"}; do
  opts=""; depth=2
  for kv in ${cfg//,/ }; do case $kv in D=*) depth=${kv#D=};; *) opts="$opts --opt $kv";; esac; done
  for rows in ${VF_SWEEP_ROWS:-1250000 1000000}; do
    echo "== $cfg rows=$rows" >> gpurun_out/split_sweep.log
    VF_BENCH_DEPTH=$depth VF_BENCH_LAUNCH=1 VF_BENCH_FORCE_EXCHANGE=1 timeout -k 10 200 python3 bench.py --gpus 1 --rows $rows --steps 400 --warmup 40 --no-cpu-baseline --no-rerank $opts 2>/dev/null \
      | python3 -c "import sys,json; [print({k: (d[k] if k!='roofline' else {kk: d[k][kk] for kk in ('frac','avg_launch_ms','pipeline_ms_per_batch')}) for k in ('ms_per_step','value','roofline','search_stats')}) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" >> gpurun_out/split_sweep.log 2>&1 || exit 1
  done
done
cat gpurun_out/split_sweep.log
