#!/bin/bash
# k_scan2 (whole-line LDS-DMA loads): parity, then A/B against k_scan, with and without the CU split
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -x -p no:cacheprovider -k "${VF_K:-fused or golden or sharding or device_and or threads or certificate or few_queries or c2_full or drop_in}" > gpurun_out/pytest_scan2.log 2>&1; rc=$?
tail -5 gpurun_out/pytest_scan2.log
if [ $rc -ne 0 ]; then grep -a "Error\|error\|assert" gpurun_out/pytest_scan2.log | head -20; exit $rc; fi
: > gpurun_out/scan2_sweep.log
run() {  # opts rows steps
  echo "== opts=[$1] rows=$2" >> gpurun_out/scan2_sweep.log
  opts=""; for kv in $1; do opts="$opts --opt $kv"; done
  VF_BENCH_DEPTH=${D:-2} VF_BENCH_LAUNCH=1 VF_BENCH_FORCE_EXCHANGE=1 timeout -k 10 200 python3 bench.py --gpus 1 --rows $2 --steps $3 --warmup 30 --no-cpu-baseline --no-rerank $opts 2>/dev/null \
    | python3 -c "import sys,json; [print({k: (d[k] if k!='roofline' else {kk: d[k][kk] for kk in ('frac','avg_launch_ms')}) for k in ('ms_per_step','roofline')}, d['search_stats']['candidates_per_query'], d['search_stats']['exact_reruns_last_batch']) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" >> gpurun_out/scan2_sweep.log 2>&1 || exit 1
}
for rows in 1000000 1250000 10000000; do
  steps=300; [ $rows -gt 5000000 ] && steps=60
  run "scan_impl=1" $rows $steps
  run "scan_impl=2" $rows $steps
  run "scan_impl=2 aux_cus=32" $rows $steps
  run "scan_impl=2 aux_cus=32 overlap_scans=1" $rows $steps
  run "scan_impl=2 overlap_scans=1" $rows $steps
done
cat gpurun_out/scan2_sweep.log
