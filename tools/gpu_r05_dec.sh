#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r05_decoder.log
timeout -k 10 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider -x -k "decoder or gemma or qwen or long_seq or llm or streaming" > $L 2>&1; rc=$?
tail -3 $L
[ $rc -ne 0 ] && tail -50 $L && exit $rc
timeout -k 10 300 python tools/bench_decoder.py 2>/dev/null | tail -4 | tee -a $L
