#!/bin/bash
# round 4, first GPU pass: smoke, the whole GPU suite (new: bench N>1 paths, full-depth decoders), the driver's bench command
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/decoder_errors.jsonl
echo "== smoke"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; rc=$?
tail -4 gpurun_out/smoke.log
[ $rc -ne 0 ] && exit $rc
echo "== pytest -m gpu"
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -40 gpurun_out/pytest_gpu.log
[ $rc -ge 124 ] && exit $rc
echo "== bench (driver command)"
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.log 2> gpurun_out/bench_driver.err; rc2=$?
tail -c 3000 gpurun_out/bench_driver.log
[ $rc2 -ne 0 ] && tail -20 gpurun_out/bench_driver.err
[ $rc -ne 0 ] && exit $rc
exit $rc2
