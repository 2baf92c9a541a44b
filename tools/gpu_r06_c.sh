#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_stamps_gap_variants.log
: > $L
run() { echo "== $*" >> $L; timeout -k 10 200 env "$@" python3 tools/stamps_gap.py 1250000 ${OPTS} >> $L 2>&1 || { tail -20 $L; exit 1; }; }
OPTS="" run A=1
OPTS="overlap_scans=0" run A=1
OPTS="" run VF_DBG_EXTRA=4
OPTS="aux_cus=0 overlap_scans=1" run A=1
OPTS="sample_rows=4" run A=1
grep -v amdgpu.ids $L
