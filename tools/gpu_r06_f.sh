#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_two_pass.log
: > $L
export VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_exp.so
echo "== isolated launches (stamps_tiles), one pass" >> $L; timeout -k 10 200 python3 tools/stamps_tiles.py 1250000 >> $L 2>&1 || { tail -20 $L; exit 1; }
echo "== isolated launches, two passes over the range in one workgroup life (debug bit 6)" >> $L; VF_DBG_EXTRA=64 timeout -k 10 200 python3 tools/stamps_tiles.py 1250000 >> $L 2>&1 || { tail -20 $L; exit 1; }
echo "== pipelined, two passes" >> $L; VF_DBG_EXTRA=64 timeout -k 10 200 python3 tools/stamps_gap.py 1250000 >> $L 2>&1 || { tail -20 $L; exit 1; }
echo "== isolated, 10M rows, one pass" >> $L; timeout -k 10 200 python3 tools/stamps_tiles.py 10000000 >> $L 2>&1 || { tail -20 $L; exit 1; }
unset VF_LIB_PATH
grep -v amdgpu.ids $L
timeout -k 10 600 python3 -m pytest tests/test_gpu_retrieval.py -k "fp8_corpus_of or small_path or fused_path_bit_exact or scan_kernels_agree" -x -q -m gpu > gpurun_out/r06_f_tests.log 2>&1 || { tail -40 gpurun_out/r06_f_tests.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r06_f_tests.log
