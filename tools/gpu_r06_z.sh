#!/bin/bash
# after the filter rewrite + the e4m3 rule: cycle accounting again, the whole GPU suite, the default bench line
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_scan2r_cycle_accounting_after.log
: > $L
for spec in "10000000 768 f16" "1250000 768 f16" "10000000 768 fp8" "10000000 1024 fp8"; do
  set -- $spec
  VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_test.so VF_DBG_EXTRA=4096 timeout -k 10 200 python3 tools/stamps_scan2r.py $1 $2 $3 >> $L 2>gpurun_out/_st.err || { tail -5 gpurun_out/_st.err; exit 1; }
done
grep -v "time per tile" $L
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r06_z_suite.log 2>&1 || { tail -30 gpurun_out/r06_z_suite.log; exit 1; }
tail -3 gpurun_out/r06_z_suite.log
