#!/bin/bash
# after the k_scan2r changes (accumulator-register B fragments, e4m3 shapes): the retrieval tests, the hook tests' child, a fuzz soak
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_retrieval.py tests/test_gpu_hooks.py -m gpu -x -q > gpurun_out/r06_v_tests.log 2>&1 || { tail -30 gpurun_out/r06_v_tests.log; exit 1; }
tail -3 gpurun_out/r06_v_tests.log
VF_FUZZ_SCAN2R=1 timeout -k 10 400 python3 tools/fuzz_search.py --seconds 240 --seed 71 > gpurun_out/r06_fuzz_scan2r_seed71.log 2>&1 || { tail -20 gpurun_out/r06_fuzz_scan2r_seed71.log; exit 1; }
tail -2 gpurun_out/r06_fuzz_scan2r_seed71.log
timeout -k 10 300 python3 tools/fuzz_search.py --seconds 150 --seed 72 > gpurun_out/r06_fuzz_seed72.log 2>&1 || { tail -20 gpurun_out/r06_fuzz_seed72.log; exit 1; }
tail -2 gpurun_out/r06_fuzz_seed72.log
