#!/bin/bash
# the whole gpu suite on the shipped library, then the experiment kernels' tests on the -DVF_EXPERIMENTS build
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/decoder_errors.jsonl
timeout -k 10 1100 python -m pytest tests -m gpu -q -x -p no:cacheprovider -rs > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -12 gpurun_out/pytest_gpu.log | cut -c1-220
if [ $rc -ne 0 ]; then grep -a "Error\|error\|assert" gpurun_out/pytest_gpu.log | head -20; exit $rc; fi
VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_exp.so timeout -k 10 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x -p no:cacheprovider -k "persistent_forward or in_the_tail or gemm_kernels or attention_kernels" > gpurun_out/pytest_gpu_experiments.log 2>&1; rc=$?
tail -4 gpurun_out/pytest_gpu_experiments.log
exit $rc
