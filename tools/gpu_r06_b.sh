#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_control_flow_golden.py -x -q -m gpu > gpurun_out/r06_b_tests.log 2>&1 || { tail -30 gpurun_out/r06_b_tests.log; exit 1; }
tail -3 gpurun_out/r06_b_tests.log
timeout -k 10 200 python3 tools/stamps_gap.py 1250000 > gpurun_out/r06_stamps_gap_1250k.log 2>&1 || { tail -20 gpurun_out/r06_stamps_gap_1250k.log; exit 1; }
cat gpurun_out/r06_stamps_gap_1250k.log
timeout -k 10 200 python3 tools/stamps_gap.py 1000000 > gpurun_out/r06_stamps_gap_1m.log 2>&1 || { tail -20 gpurun_out/r06_stamps_gap_1m.log; exit 1; }
cat gpurun_out/r06_stamps_gap_1m.log
