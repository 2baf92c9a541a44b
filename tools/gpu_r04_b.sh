#!/bin/bash
# round 4, second pass: the new tests only (CLIP text tower, mixed-modality index, cosine rows, full-depth decoders), then the bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_vision.py tests/test_gpu_retrieval.py tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider -x \
  -k "text_tower or clip_text or mixed_modality or cosine_matrix_of_index_rows or full_depth or reranker_matches_torch" -s > gpurun_out/pytest_new.log 2>&1; rc=$?
grep -E "text tower|measured|passed|failed|Error|error" gpurun_out/pytest_new.log | tail -30
[ $rc -ne 0 ] && tail -60 gpurun_out/pytest_new.log && exit $rc
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.log 2> gpurun_out/bench_driver.err; rc2=$?
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_driver.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(json.dumps(d.get('c4'), indent=0)[:3000])
PY
[ $rc2 -ne 0 ] && tail -20 gpurun_out/bench_driver.err
exit $rc2
