#!/bin/bash
# round 3 record pass: whole gpu suite, the driver's bench command, the self-launched 1.25M-row rehearsal with the exchange and
# the oracle verification, configs[1], and rocprofv3 kernel stats of the small-shard and 10M workloads (ordered scans)
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/decoder_errors.jsonl
echo "== pytest -m gpu (${VF_K:-all})"
timeout -k 10 1100 python -m pytest tests -m gpu -q -x -p no:cacheprovider ${VF_K:+-k "$VF_K"} > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -6 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then grep -a "Error\|error\|assert" gpurun_out/pytest_gpu.log | head -20; exit $rc; fi
echo "== bench (driver command)"
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_cmd.log 2> gpurun_out/bench_driver_cmd.err; rc=$?
tail -c 1500 gpurun_out/bench_driver_cmd.log; tail -3 gpurun_out/bench_driver_cmd.err
if [ $rc -ne 0 ]; then exit $rc; fi
echo "== rehearsal 1.25M, self-launched, exchange forced, verified"
VF_BENCH_LAUNCH=1 VF_BENCH_FORCE_EXCHANGE=1 timeout -k 10 400 python3 bench.py --gpus 1 --rows 1250000 --steps 400 --warmup 40 --no-cpu-baseline --no-rerank --verify > gpurun_out/bench_rehearsal_1250k.log 2> gpurun_out/bench_rehearsal_1250k.err; rc=$?
tail -c 2500 gpurun_out/bench_rehearsal_1250k.log; grep -a "verify" gpurun_out/bench_rehearsal_1250k.err | tail -2
if [ $rc -ne 0 ]; then tail -5 gpurun_out/bench_rehearsal_1250k.err; exit $rc; fi
echo "== configs[1]: 1M rows"
timeout -k 10 300 python3 bench.py --gpus 1 --rows 1000000 --steps 400 --warmup 40 --no-rerank > gpurun_out/bench_c2_1m.log 2>/dev/null; rc=$?
tail -c 2500 gpurun_out/bench_c2_1m.log
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "1250000 200 r03_kernel_stats_1250k" "10000000 40 r03_kernel_stats_10Mx768"; do
  set -- $cfg
  rm -rf /tmp/prof_$3
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$3 -o out -- python3 $R/bench.py --gpus 1 --rows $1 --steps $2 --warmup 10 --no-cpu-baseline --no-rerank > $R/gpurun_out/$3.bench.log 2>/dev/null
  f=$(find /tmp/prof_$3 -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -12 "$f" | cut -c1-220 > $R/gpurun_out/$3.csv
  t=$(find /tmp/prof_$3 -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 $R/tools/trace_span.py "$t" k_scan2 15 > $R/gpurun_out/$3.span.txt
  python3 -c "import sys,json; d=[json.loads(l) for l in open('$R/gpurun_out/$3.bench.log') if l.startswith('{')][0]; print(d['ms_per_step'], json.dumps(d['roofline'])[:900])"
  head -5 $R/gpurun_out/$3.csv | cut -c1-150; cat $R/gpurun_out/$3.span.txt
done
