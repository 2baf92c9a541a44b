#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_cu_mapped.log
echo "#### second form: the table CU -> range is filled by the first launch (a bijection in isolated launches)" >> $L
export VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_exp.so
echo "== isolated launches, ranges by blockIdx" >> $L; timeout -k 10 200 python3 tools/stamps_tiles.py 1250000 >> $L 2>&1 || { tail -20 $L; exit 1; }
echo "== isolated launches, ranges by compute unit (debug bit 5)" >> $L; VF_DBG_EXTRA=32 timeout -k 10 200 python3 tools/stamps_tiles.py 1250000 >> $L 2>&1 || { tail -20 $L; exit 1; }
echo "== isolated launches, ranges by compute unit, candidate path off" >> $L; VF_DBG_EXTRA=36 timeout -k 10 200 python3 tools/stamps_tiles.py 1250000 >> $L 2>&1 || { tail -20 $L; exit 1; }
grep -v amdgpu.ids $L | tail -60
