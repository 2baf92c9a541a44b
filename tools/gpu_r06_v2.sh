#!/bin/bash
# soak after the filter / candidate-path rewrite: fuzz (general + k_scan2r shapes), the 40-run stress
set -o pipefail
mkdir -p gpurun_out
VF_FUZZ_SCAN2R=1 timeout -k 10 400 python3 tools/fuzz_search.py --seconds 200 --seed 81 > gpurun_out/r06_fuzz_scan2r_seed81.log 2>&1 || { tail -20 gpurun_out/r06_fuzz_scan2r_seed81.log; exit 1; }
tail -1 gpurun_out/r06_fuzz_scan2r_seed81.log
timeout -k 10 400 python3 tools/fuzz_search.py --seconds 240 --seed 82 > gpurun_out/r06_fuzz_seed82.log 2>&1 || { tail -20 gpurun_out/r06_fuzz_seed82.log; exit 1; }
tail -1 gpurun_out/r06_fuzz_seed82.log
timeout -k 10 400 python3 tools/fuzz_search.py --seconds 120 --seed 83 --repeat 4 > gpurun_out/r06_fuzz_seed83_repeat4.log 2>&1 || { tail -20 gpurun_out/r06_fuzz_seed83_repeat4.log; exit 1; }
tail -1 gpurun_out/r06_fuzz_seed83_repeat4.log
timeout -k 10 300 python3 tools/stress_repeat.py --runs 40 > gpurun_out/r06_stress_repeat_40_after_filter.log 2>&1 || { tail -20 gpurun_out/r06_stress_repeat_40_after_filter.log; exit 1; }
tail -3 gpurun_out/r06_stress_repeat_40_after_filter.log
