#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O2 -o /tmp/cu_mask_probe tools/ubench/cu_mask_probe.hip 2>/dev/null && timeout -k 5 60 /tmp/cu_mask_probe > gpurun_out/cu_mask_probe.log 2>&1
cat gpurun_out/cu_mask_probe.log
timeout -k 10 600 python -m pytest tests/test_gpu_retrieval.py -m gpu -q -x -p no:cacheprovider -k "few_queries or c4_5m" -s > gpurun_out/pytest_c4.log 2>&1; rc=$?
grep -a "stats\|passed\|failed\|Error" gpurun_out/pytest_c4.log | tail -12
exit $rc
