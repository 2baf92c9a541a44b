#!/bin/bash
# One GPU-box pass: smoke, parity tests, short benches.  Usage: gpurun -- bash tools/gpu_check.sh
set -o pipefail
mkdir -p gpurun_out
echo "== smoke" | tee gpurun_out/smoke.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/smoke.log 2>&1; rc=$?
tail -5 gpurun_out/smoke.log
if [ $rc -ge 124 ]; then echo "smoke timed out/killed rc=$rc"; exit $rc; fi
echo "== pytest -m gpu"
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -25 gpurun_out/pytest_gpu.log
if [ $rc -ge 124 ]; then echo "pytest timed out/killed rc=$rc"; exit $rc; fi
echo "== bench 1M"
timeout -k 10 300 python bench.py --rows 1000000 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_1m.log 2>&1; rc=$?
tail -3 gpurun_out/bench_1m.log
if [ $rc -ge 124 ]; then echo "bench timed out rc=$rc"; exit $rc; fi
echo "== bench 10M"
timeout -k 10 420 python bench.py --steps 50 --warmup 5 > gpurun_out/bench_10m.log 2>&1; rc=$?
tail -3 gpurun_out/bench_10m.log
exit 0
