#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_scan2r_sizes.log
: > $L
export R06_CFGS='[{"scan_impl":4},{"scan_impl":5}]'
for rows in 2500000 5000000 1250000; do
  R06_REPS=3 timeout -k 10 300 python3 tools/r06_small_sweep.py $rows 768 >> $L 2>&1 || { tail -20 $L; exit 1; }
done
grep -v amdgpu.ids $L
t0=$(date +%s)
VF_TEST_FUZZ_SECONDS=60 timeout -k 10 1100 python3 -m pytest tests/ -x -q -m gpu --durations=8 > gpurun_out/r06_suite.log 2>&1
rc=$?
echo "suite rc $rc wall $(( $(date +%s) - t0 )) s"
tail -16 gpurun_out/r06_suite.log | cut -c1-200
exit $rc
