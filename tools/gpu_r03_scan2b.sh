#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -x -p no:cacheprovider -k "scan_kernels or fused or golden or sharding or device_and or certificate or few_queries or c2_full" > gpurun_out/pytest_scan2.log 2>&1; rc=$?
tail -3 gpurun_out/pytest_scan2.log
if [ $rc -ne 0 ]; then grep -a "Error\|error\|assert" gpurun_out/pytest_scan2.log | head -20; exit $rc; fi
: > gpurun_out/scan2_sweep3.log
run() {  # depth opts rows steps
  echo "== D=$1 opts=[$2] rows=$3" >> gpurun_out/scan2_sweep3.log
  opts=""; for kv in $2; do opts="$opts --opt $kv"; done
  VF_BENCH_DEPTH=$1 VF_BENCH_LAUNCH=1 VF_BENCH_FORCE_EXCHANGE=1 timeout -k 10 200 python3 bench.py --gpus 1 --rows $3 --steps $4 --warmup 30 --no-cpu-baseline --no-rerank $opts 2>/dev/null \
    | python3 -c "import sys,json; [print(d['ms_per_step'], {kk: d['roofline'].get(kk) for kk in ('frac','avg_launch_ms')}, (d['roofline'].get('isolated_launch') or {}).get('avg_launch_ms'), d['search_stats']['candidates_per_query'], d['search_stats']['exact_reruns_last_batch']) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" >> gpurun_out/scan2_sweep3.log 2>&1 || exit 1
}
for rows in 1000000 1250000 2500000 10000000; do
  steps=300; [ $rows -gt 2000000 ] && steps=100
  run 2 "" $rows $steps
  run 2 "aux_cus=0 overlap_scans=0" $rows $steps
  run 2 "scan_impl=1 aux_cus=0 overlap_scans=0" $rows $steps
done
cat gpurun_out/scan2_sweep3.log
