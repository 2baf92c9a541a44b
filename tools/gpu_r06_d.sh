#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_rotation.log
: > $L
echo "== stamps, default order" >> $L; timeout -k 10 200 python3 tools/stamps_gap.py 1250000 >> $L 2>&1 || { tail -20 $L; exit 1; }
echo "== stamps, rotated ranges (debug bit 4)" >> $L; VF_DBG_EXTRA=16 timeout -k 10 200 python3 tools/stamps_gap.py 1250000 >> $L 2>&1 || { tail -20 $L; exit 1; }
export R06_CFGS='[{"aux_cus":32,"sample_rows":16},{"aux_cus":32,"sample_rows":16,"debug":16},{"aux_cus":32,"sample_rows":8},{"aux_cus":32,"sample_rows":8,"debug":16},{"aux_cus":32,"sample_rows":16,"debug":20},{"aux_cus":32,"sample_rows":16,"debug":4}]'
R06_REPS=3 timeout -k 10 300 python3 tools/r06_small_sweep.py 1250000 768 >> $L 2>&1 || { tail -20 $L; exit 1; }
R06_REPS=2 timeout -k 10 300 python3 tools/r06_small_sweep.py 1000000 768 >> $L 2>&1 || { tail -20 $L; exit 1; }
export R06_CFGS='[{"aux_cus":0},{"aux_cus":0,"debug":16}]'
R06_REPS=2 timeout -k 10 300 python3 tools/r06_small_sweep.py 10000000 768 >> $L 2>&1 || { tail -20 $L; exit 1; }
grep -v amdgpu.ids $L
timeout -k 10 600 python3 -m pytest tests/test_gpu_encoder.py tests/test_pretrained.py tests/test_control_flow_golden.py -x -q -m gpu > gpurun_out/r06_d_tests.log 2>&1 || { tail -30 gpurun_out/r06_d_tests.log; exit 1; }
tail -3 gpurun_out/r06_d_tests.log
