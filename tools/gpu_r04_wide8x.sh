#!/bin/bash
# timing experiments of k_scan_wide8 at the 8-GPU shard size: phase stamps of the shipped build and of two variants with invalid results
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_wide8x.log
: > $L
for v in "" nocand noepi; do
  echo "== variant '$v'" | tee -a $L
  if [ -n "$v" ]; then export VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_$v.so; fi
  timeout -k 10 300 python tools/stamps_wide8.py 1250000 >> $L 2>&1 || { tail -20 $L; exit 1; }
done
grep -v amdgpu.ids $L
