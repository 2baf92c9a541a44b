#!/bin/bash
# round 3, first GPU pass: new/changed tests first (fast feedback), then the driver's bench command, then the self-launched
# one-rank rehearsal of the N>1 path (RCCL world 1, exchange + oracle verification), then the whole gpu suite.
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/decoder_errors.jsonl
echo "== new tests"
timeout -k 10 900 python -m pytest tests/test_control_flow_golden.py tests/test_gpu_encoder.py -m gpu -q -x -p no:cacheprovider -k "g5 or g6 or decoder or gemma or llm_reranker" > gpurun_out/pytest_new.log 2>&1; rc=$?
tail -15 gpurun_out/pytest_new.log
if [ $rc -ge 124 ]; then exit $rc; fi
echo "== bench (driver command)"
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver_cmd.log 2> gpurun_out/bench_driver_cmd.err; rc=$?
tail -c 6000 gpurun_out/bench_driver_cmd.log; tail -5 gpurun_out/bench_driver_cmd.err
if [ $rc -ge 124 ]; then exit $rc; fi
echo "== self-launched one-rank rehearsal (1.25M rows, exchange forced)"
VF_BENCH_LAUNCH=1 VF_BENCH_FORCE_EXCHANGE=1 timeout -k 10 400 python3 bench.py --gpus 1 --rows 1250000 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_rehearsal.log 2> gpurun_out/bench_rehearsal.err; rc=$?
tail -c 3000 gpurun_out/bench_rehearsal.log; tail -5 gpurun_out/bench_rehearsal.err
if [ $rc -ge 124 ]; then exit $rc; fi
echo "== pytest -m gpu (all)"
timeout -k 10 1100 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -15 gpurun_out/pytest_gpu.log
exit $rc
