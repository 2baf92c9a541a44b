#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_gpu_retrieval.py -k "scan2r or scan_kernels_agree_and_match" -x -q -m gpu > gpurun_out/r06_h_tests.log 2>&1 || { tail -40 gpurun_out/r06_h_tests.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r06_h_tests.log
L=gpurun_out/r06_scan2r.log
: > $L
export R06_CFGS='[{"aux_cus":32},{"aux_cus":32,"scan_impl":5},{"aux_cus":32,"debug":4},{"aux_cus":32,"scan_impl":5,"debug":4}]'
R06_REPS=3 timeout -k 10 300 python3 tools/r06_small_sweep.py 1250000 768 >> $L 2>&1 || { tail -20 $L; exit 1; }
R06_REPS=2 timeout -k 10 300 python3 tools/r06_small_sweep.py 1000000 768 >> $L 2>&1 || { tail -20 $L; exit 1; }
export R06_CFGS='[{"aux_cus":0},{"aux_cus":0,"scan_impl":5}]'
R06_REPS=2 timeout -k 10 300 python3 tools/r06_small_sweep.py 10000000 768 >> $L 2>&1 || { tail -20 $L; exit 1; }
grep -v amdgpu.ids $L
echo "== stamps, k_scan2r pipelined" >> $L; timeout -k 10 200 python3 tools/stamps_gap.py 1250000 scan_impl=5 >> $L 2>&1 || { tail -20 $L; exit 1; }
grep -v amdgpu.ids $L | tail -22
