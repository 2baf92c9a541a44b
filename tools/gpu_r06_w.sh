#!/bin/bash
# k_scan2r: a wave's cycles by phase (test variant, debug bit 12)
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_scan2r_cycle_accounting.log
: > $L
export VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_test.so
for spec in "10000000 768 f16" "1250000 768 f16" "10000000 768 fp8" "10000000 1024 fp8"; do
  set -- $spec
  VF_DBG_EXTRA=4096 timeout -k 10 200 python3 tools/stamps_scan2r.py $1 $2 $3 >> $L 2>gpurun_out/_st.err || { tail -5 gpurun_out/_st.err; exit 1; }
done
cat $L
