#!/bin/bash
# soak of the final tree (non-temporal row loads): fuzz with and without the k_scan2r shapes, repeated runs, the 40-run stress
set -o pipefail
mkdir -p gpurun_out
VF_FUZZ_SCAN2R=1 timeout -k 10 400 python3 tools/fuzz_search.py --seconds 180 --seed 111 > gpurun_out/r06_fuzz_scan2r_seed111.log 2>&1 || { tail -20 gpurun_out/r06_fuzz_scan2r_seed111.log; exit 1; }
tail -1 gpurun_out/r06_fuzz_scan2r_seed111.log
timeout -k 10 400 python3 tools/fuzz_search.py --seconds 180 --seed 112 > gpurun_out/r06_fuzz_seed112.log 2>&1 || { tail -20 gpurun_out/r06_fuzz_seed112.log; exit 1; }
tail -1 gpurun_out/r06_fuzz_seed112.log
VF_FUZZ_SCAN2R=1 timeout -k 10 300 python3 tools/fuzz_search.py --seconds 90 --seed 113 --repeat 4 > gpurun_out/r06_fuzz_scan2r_seed113_repeat4.log 2>&1 || { tail -20 gpurun_out/r06_fuzz_scan2r_seed113_repeat4.log; exit 1; }
tail -1 gpurun_out/r06_fuzz_scan2r_seed113_repeat4.log
timeout -k 10 300 python3 tools/stress_repeat.py --runs 40 > gpurun_out/r06_stress_repeat_40_nt.log 2>&1 || { tail -20 gpurun_out/r06_stress_repeat_40_nt.log; exit 1; }
echo "stress cases with 0 failures: $(grep -c '"failures": 0' gpurun_out/r06_stress_repeat_40_nt.log) of $(grep -c '"runs"' gpurun_out/r06_stress_repeat_40_nt.log)"
