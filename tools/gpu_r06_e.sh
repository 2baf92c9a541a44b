#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_stagger.log
: > $L
echo "== stamps, waves of a workgroup staggered (debug bit 5)" >> $L; VF_DBG_EXTRA=32 timeout -k 10 200 python3 tools/stamps_gap.py 1250000 >> $L 2>&1 || { tail -20 $L; exit 1; }
export R06_CFGS='[{"aux_cus":32,"sample_rows":16},{"aux_cus":32,"sample_rows":16,"debug":32},{"aux_cus":32,"sample_rows":16,"debug":36},{"aux_cus":32,"sample_rows":16,"debug":4}]'
R06_REPS=3 timeout -k 10 300 python3 tools/r06_small_sweep.py 1250000 768 >> $L 2>&1 || { tail -20 $L; exit 1; }
grep -v amdgpu.ids $L
timeout -k 10 600 python3 -m pytest tests/test_pretrained.py tests/test_gpu_retrieval.py -k "pretrained or from_config or fp8_corpus_of or embedder or pair_inputs or reranker or drop_in" -x -q -m gpu > gpurun_out/r06_e_tests.log 2>&1 || { tail -40 gpurun_out/r06_e_tests.log | cut -c1-300; exit 1; }
tail -3 gpurun_out/r06_e_tests.log
