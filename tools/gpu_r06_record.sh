#!/bin/bash
# Round 6 record: the driver's command (whole line), kernel stats + launch-interval spans of the headline, configs[1], the 8-GPU rank's
# shard (RCCL exchange in the loop) and configs[4], PMC traffic of the three narrow-scan workloads.  Usage: gpurun -- bash tools/gpu_r06_record.sh [a|b]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
part=${1:-a}
stats() {   # tag, kernel substring for the span, bench args...
  tag=$1; needle=$2; shift 2
  mkdir -p $R/gpurun_out/$tag
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -o $tag -- python3 $R/bench.py "$@" > $R/gpurun_out/$tag/run.log 2>&1 ) || { tail -3 gpurun_out/$tag/run.log; return 1; }
  f=$(find gpurun_out/$tag -name "*kernel_stats.csv" | head -1); t=$(find gpurun_out/$tag -name "*kernel_trace.csv" | head -1)
  cp $f gpurun_out/${tag}_kernel_stats.csv
  head -7 $f | cut -c1-150
  python3 tools/trace_span.py $t "$needle" ${SKIP:-20} ${COUNT:-200} | tee gpurun_out/${tag}_span.txt
  grep '^{' gpurun_out/$tag/run.log | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); r=j['roofline']; print('the bench line of this run (under the profiler): ms_per_step', j['ms_per_step'], 'roofline.frac', r['frac'], 'avg_launch_ms', r.get('avg_launch_ms'))" | tee -a gpurun_out/${tag}_span.txt
  find gpurun_out/$tag -name "*.csv" -delete
}
pmc() {   # tag, counter, bench args...
  tag=$1; ctr=$2; shift 2
  mkdir -p $R/gpurun_out/$tag
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $R/gpurun_out/$tag -o $tag -- python3 $R/bench.py "$@" > $R/gpurun_out/$tag/run.log 2>&1 ) || { tail -3 gpurun_out/$tag/run.log; return 1; }
  f=$(find gpurun_out/$tag -name "*counter_collection.csv" | head -1)
  cp $f gpurun_out/${tag}_${ctr}.csv
  grep '^{' gpurun_out/$tag/run.log | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['roofline']['bytes_per_launch'])" > gpurun_out/${tag}_alg_bytes.txt
  find gpurun_out/$tag -name "*.csv" -delete
}
if [ "$part" = "a" ]; then
  t0=$(date +%s)
  timeout -k 10 1000 python3 -m pytest tests/ -x -q -m gpu > gpurun_out/r06_suite.log 2>&1; rc=$?
  echo "suite rc $rc wall $(( $(date +%s) - t0 )) s: $(tail -1 gpurun_out/r06_suite.log)"
  [ $rc -eq 0 ] || { tail -30 gpurun_out/r06_suite.log | cut -c1-200; exit $rc; }
  bash tools/gpu_r06_bench.sh || exit 1
else
  N="--no-cpu-baseline --no-rerank --no-shard-legs --no-startup"
  export VF_BENCH_NO_ISOLATED=1      # the traces hold the pipelined loop only
  echo "== kernel stats, headline 10M x 768"; SKIP=5 COUNT=20 stats r06_10m "k_scan2r<2, 1, 0" --gpus 1 --steps 20 --warmup 5 $N || exit 1
  echo "== kernel stats, headline, ORDERED scans (one launch at a time on the scan partition: what roofline.isolated_launch times)"; SKIP=5 COUNT=20 stats r06_10m_ordered "k_scan2r<2, 1, 0" --gpus 1 --steps 20 --warmup 5 $N --opt overlap_scans=0 --opt scan_impl=5 || exit 1
  [ "$2" = "ordered-only" ] && exit 0
  echo "== kernel stats, configs[1] 1M x 768"; stats r06_c2 "k_scan2<2, 0>" --gpus 1 --rows 1000000 --steps 200 --warmup 20 $N || exit 1
  echo "== kernel stats, the 8-GPU rank's shard 1.25M x 768 with the RCCL exchange in the loop (one rank, no launcher: rocprofv3 does not follow one)"
  export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 VF_BENCH_FORCE_EXCHANGE=1
  stats r06_shard8 "k_scan2r<2, 1, 0" --gpus 1 --rows 1250000 --steps 200 --warmup 20 --verify $N || exit 1
  unset RANK LOCAL_RANK WORLD_SIZE MASTER_ADDR MASTER_PORT VF_BENCH_FORCE_EXCHANGE
  C5="--rows 10000000 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --no-startup"
  echo "== kernel stats, configs[4]"; SKIP=3 COUNT=24 stats r06_c5 "k_scan_wide8" $C5 --steps 24 --warmup 3 || exit 1
  [ "$2" = "stats-only" ] && exit 0
  S="--steps 12 --warmup 3 $N --opt overlap_scans=0"
  # (ordered scans for the counter passes: one launch per bracket; k_scan2r is the default only where scans overlap, so it is named)
  for w in "10m 10000000 5" "c2 1000000 2" "s8 1250000 5"; do
    set -- $w
    echo "== PMC $1"; pmc r06_${1}_fetch FETCH_SIZE --gpus 1 --rows $2 $S --opt scan_impl=$3 || exit 1
    pmc r06_${1}_write WRITE_SIZE --gpus 1 --rows $2 $S --opt scan_impl=$3 || exit 1
  done
  python3 tools/pmc_traffic.py gpurun_out/r06_10m_fetch_FETCH_SIZE.csv gpurun_out/r06_10m_write_WRITE_SIZE.csv "k_scan2r<2, 1, 0" 10000000 768 64 100 $(cat gpurun_out/r06_10m_fetch_alg_bytes.txt) gpurun_out/pmc_traffic_scan2_10Mx768.json "round 6 (tools/gpu_r06_record.sh b)"
  python3 tools/pmc_traffic.py gpurun_out/r06_c2_fetch_FETCH_SIZE.csv gpurun_out/r06_c2_write_WRITE_SIZE.csv "k_scan2<2, 0>" 1000000 768 64 100 $(cat gpurun_out/r06_c2_fetch_alg_bytes.txt) gpurun_out/pmc_traffic_scan2_1000k.json "round 6 (tools/gpu_r06_record.sh b)"
  python3 tools/pmc_traffic.py gpurun_out/r06_s8_fetch_FETCH_SIZE.csv gpurun_out/r06_s8_write_WRITE_SIZE.csv "k_scan2r<2, 1, 0" 1250000 768 64 100 $(cat gpurun_out/r06_s8_fetch_alg_bytes.txt) gpurun_out/pmc_traffic_scan2_1250k.json "round 6 (tools/gpu_r06_record.sh b)"
  rm -f gpurun_out/r06_*_FETCH_SIZE.csv gpurun_out/r06_*_WRITE_SIZE.csv
fi
