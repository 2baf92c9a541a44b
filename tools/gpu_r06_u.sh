#!/bin/bash
# k_scan2r: where does a tile's time go, and at what clock?  (test variant; experiment bits give INVALID results, timing only)
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_scan2r_stamps_clock.log
: > $L
export VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_test.so
for spec in "10000000 768 fp8 0" "10000000 768 fp8 32" "10000000 768 fp8 4" "10000000 768 f16 0" "10000000 768 f16 4" "10000000 1024 fp8 0"; do
  set -- $spec
  VF_DBG_EXTRA=$4 timeout -k 10 200 python3 tools/stamps_scan2r.py $1 $2 $3 >> $L 2>gpurun_out/_st.err || { tail -5 gpurun_out/_st.err; exit 1; }
done
cat $L
