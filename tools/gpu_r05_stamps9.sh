#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r05_split9_stamps.log
timeout -k 10 300 python tools/gemm9_split_stamps.py --epi 2 2>&1 | grep -v amdgpu.ids | tee $L
timeout -k 10 300 python tools/gemm9_split_stamps.py --epi 0 2>&1 | grep -v amdgpu.ids | tee -a $L
timeout -k 10 600 python -m pytest tests/test_vision.py tests/test_pretrained.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
