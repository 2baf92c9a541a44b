#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -x -p no:cacheprovider -k "fp8 or e4m3 or scan_kernels" > gpurun_out/pytest_f8.log 2>&1; rc=$?
tail -3 gpurun_out/pytest_f8.log
if [ $rc -ne 0 ]; then grep -a "Error\|error\|assert" gpurun_out/pytest_f8.log | head -20; exit $rc; fi
: > gpurun_out/f8_sweep.log
for cfg in "10000000 768 60" "10000000 1024 60" "1250000 768 300"; do
  set -- $cfg
  for impl in 1 2 1 2; do
    echo "== fp8 rows=$1 d=$2 scan_impl=$impl" >> gpurun_out/f8_sweep.log
    timeout -k 10 200 python3 bench.py --gpus 1 --rows $1 --dim $2 --corpus-dtype fp8 --steps $3 --warmup 20 --no-cpu-baseline --no-rerank --opt scan_impl=$impl 2>/dev/null \
      | python3 -c "import sys,json; [print(d['ms_per_step'], d['value'], {kk: d['roofline'].get(kk) for kk in ('frac','avg_launch_ms','kernel')}) for d in [json.loads(l) for l in sys.stdin if l.startswith('{')]]" >> gpurun_out/f8_sweep.log 2>&1 || exit 1
  done
done
cat gpurun_out/f8_sweep.log
