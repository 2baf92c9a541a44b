#!/bin/bash
# round 5: where the waves of k_scan_wide8 spend a launch, 8-wave and 4-wave form (stamps build)
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r05_stamps.log
: > $L
for w in 8 4; do
  for rows in 10000000; do
    echo "== wide8_waves=$w rows $rows" | tee -a $L
    VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_stamps.so timeout -k 10 300 python tools/stamps_wide8.py $rows wide8_waves=$w 2>&1 | grep -v amdgpu.ids | tee -a $L
  done
done
