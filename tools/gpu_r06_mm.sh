#!/bin/bash
# after the base-row clamp: the regression test, then the soak that found the fault (same seeds), one pass
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_retrieval.py -m gpu -x -q -k "shorter_than_their_sample_part or scan2r" > gpurun_out/r06_mm_tests.log 2>&1 || { tail -30 gpurun_out/r06_mm_tests.log; exit 1; }
tail -2 gpurun_out/r06_mm_tests.log
bash tools/gpu_r06_ll.sh
