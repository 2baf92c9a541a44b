#!/bin/bash
# (the A/B partner is built here first and travels with the snapshot:
#   VF_BUILD_FLAGS="-DVF_SCAN2_SERVICE=1" VF_BUILD_LIB=libvf_nosvc.so VF_BUILD_TAG=_nosvc python -m veritasfi_amd.build
#  -- the name dates from the run in which the service wave was the default build and this library the four-wave form)
# A/B of k_scan2's service wave (VF_SCAN2_SERVICE=0 build in lib/libvf_nosvc.so): parity first, then timing at 1M..10M rows
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -x -p no:cacheprovider -k "scan_kernels or fused or golden or sharding or device_and or certificate or few_queries or c2_full" > gpurun_out/pytest_svc.log 2>&1; rc=$?
tail -3 gpurun_out/pytest_svc.log
if [ $rc -ne 0 ]; then grep -a "Error\|assert" gpurun_out/pytest_svc.log | head -20; exit $rc; fi
: > gpurun_out/svc_ab.log
run() {  # lib opts rows steps
  echo "== lib=$1 opts=[$2] rows=$3" >> gpurun_out/svc_ab.log
  opts=""; for kv in $2; do opts="$opts --opt $kv"; done
  lib=""; [ -n "$1" ] && lib="$PWD/veritasfi_amd/lib/$1"
  VF_LIB_PATH=$lib VF_BENCH_DEPTH=2 VF_BENCH_LAUNCH=1 VF_BENCH_FORCE_EXCHANGE=1 timeout -k 10 200 python3 bench.py --gpus 1 --rows $3 --steps $4 --warmup 30 --no-cpu-baseline --no-rerank $opts 2>/dev/null \
    | python3 -c "import sys,json; [print(d['ms_per_step'], {kk: d['roofline'].get(kk) for kk in ('frac','avg_launch_ms')}, (d['roofline'].get('isolated_launch') or {}).get('avg_launch_ms'), d['search_stats']['candidates_per_query'], d['search_stats']['exact_reruns_last_batch'], d.get('verified')) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" >> gpurun_out/svc_ab.log 2>&1 || exit 1
}
for rows in 1250000 2500000 10000000; do
  steps=400; [ $rows -gt 2000000 ] && steps=100
  run "" "" $rows $steps
  run libvf_nosvc.so "" $rows $steps
  run "" "" $rows $steps
  run libvf_nosvc.so "" $rows $steps
done
cat gpurun_out/svc_ab.log
timeout -k 10 120 python3 tools/stamps_tiles.py > gpurun_out/stamps_tiles_svc.log 2>&1; tail -18 gpurun_out/stamps_tiles_svc.log
