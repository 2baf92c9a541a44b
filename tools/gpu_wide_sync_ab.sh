set -o pipefail
C5="--rows 10000000 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --steps 10 --warmup 2"
timeout -k 10 500 python -m pytest tests/test_gpu_retrieval.py -x -q -m gpu -k "wide or c5" 2>&1 | tail -3 || exit 1
for s in -1 1 0 2; do
  timeout -k 10 300 python bench.py $C5 --opt wide_sync=$s > gpurun_out/wide_sync_$s.log 2>&1 || { tail -3 gpurun_out/wide_sync_$s.log; exit 1; }
  python3 - gpurun_out/wide_sync_$s.log $s <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
d = json.loads(l)
print("wide_sync", sys.argv[2], "q/s", d["value"], "ms/step", d["ms_per_step"], "scan ms", d["roofline"]["avg_launch_ms"], "TF", d["roofline"]["achieved"])
PY
done
