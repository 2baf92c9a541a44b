#!/bin/bash
# One GPU-box pass: smoke, parity tests, short benches.  Usage: gpurun -- bash tools/gpu_check.sh
# Exit code = the worst stage's (a timed-out / killed stage stops the run: no further GPU step after a hang).
set -o pipefail
mkdir -p gpurun_out
worst=0
note() { local rc=$1 what=$2; if [ "$rc" -ne 0 ]; then echo "FAILED: $what rc=$rc"; [ "$rc" -gt "$worst" ] && worst=$rc; fi; }
echo "== smoke" | tee gpurun_out/smoke.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/smoke.log 2>&1; rc=$?
tail -5 gpurun_out/smoke.log; note $rc smoke
if [ $rc -ge 124 ]; then exit $rc; fi
echo "== pytest -m gpu"
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -p no:cacheprovider ${VF_PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -25 gpurun_out/pytest_gpu.log; note $rc "pytest -m gpu"
if [ $rc -ge 124 ]; then exit $rc; fi
if [ -z "$VF_SKIP_BENCH" ]; then
echo "== bench 1M"
timeout -k 10 300 python bench.py --rows 1000000 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_1m.log 2>&1; rc=$?
tail -3 gpurun_out/bench_1m.log; note $rc "bench 1M"
if [ $rc -ge 124 ]; then exit $rc; fi
echo "== bench 10M"
timeout -k 10 420 python bench.py --steps 50 --warmup 5 > gpurun_out/bench_10m.log 2>&1; rc=$?
tail -3 gpurun_out/bench_10m.log; note $rc "bench 10M"
fi
exit $worst
