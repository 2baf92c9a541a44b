#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_scan2r_10m.log
: > $L
export R06_CFGS='[{"aux_cus":0,"scan_impl":4},{"aux_cus":0,"scan_impl":5},{"aux_cus":32,"overlap_scans":1,"scan_impl":4},{"aux_cus":32,"overlap_scans":1,"scan_impl":5},{"aux_cus":32,"overlap_scans":0,"scan_impl":5},{"aux_cus":0,"overlap_scans":1,"scan_impl":5}]'
R06_REPS=2 timeout -k 10 500 python3 tools/r06_small_sweep.py 10000000 768 >> $L 2>&1 || { tail -20 $L; exit 1; }
R06_REPS=2 timeout -k 10 500 python3 tools/r06_small_sweep.py 7500000 768 >> $L 2>&1 || { tail -20 $L; exit 1; }
grep -v amdgpu.ids $L
