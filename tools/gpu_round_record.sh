#!/bin/bash
# End-of-milestone record: smoke, all gpu tests, the default bench line, kernel stats + PMC passes of the headline configs.
# Usage: gpurun --timeout 1200 -- bash tools/gpu_round_record.sh <tag>
set -o pipefail
tag=${1:-r02}
mkdir -p gpurun_out
echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1 || { tail -5 gpurun_out/${tag}_smoke.log; exit 1; }
tail -4 gpurun_out/${tag}_smoke.log
echo "== pytest -m gpu"; timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/${tag}_pytest_gpu.log 2>&1 || { tail -15 gpurun_out/${tag}_pytest_gpu.log; exit 1; }
tail -2 gpurun_out/${tag}_pytest_gpu.log
echo "== default bench"; timeout -k 10 500 python bench.py > gpurun_out/${tag}_bench_10m.log 2>&1 || { tail -5 gpurun_out/${tag}_bench_10m.log; exit 1; }
tail -1 gpurun_out/${tag}_bench_10m.log | cut -c1-600
echo "== kernel stats, 10M x 768"; bash tools/gpu_prof_bench.sh ${tag}_10m stats --steps 50 --warmup 5 --no-cpu-baseline --no-rerank || exit 1
echo "== PMC, 10M x 768"; bash tools/gpu_prof_bench.sh ${tag}_10m_fetch FETCH_SIZE --steps 6 --warmup 2 --no-cpu-baseline --no-rerank || exit 1
bash tools/gpu_prof_bench.sh ${tag}_10m_write WRITE_SIZE --steps 6 --warmup 2 --no-cpu-baseline --no-rerank || exit 1
C5="--rows 10000000 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank"
echo "== C5 bench"; timeout -k 10 400 python bench.py $C5 --steps 10 --warmup 2 > gpurun_out/${tag}_bench_c5_10m.log 2>&1 || exit 1
tail -1 gpurun_out/${tag}_bench_c5_10m.log | cut -c1-500
echo "== kernel stats, C5"; bash tools/gpu_prof_bench.sh ${tag}_c5 stats $C5 --steps 6 --warmup 2 || exit 1
echo "== PMC, C5"; bash tools/gpu_prof_bench.sh ${tag}_c5_fetch FETCH_SIZE $C5 --steps 4 --warmup 1 || exit 1
bash tools/gpu_prof_bench.sh ${tag}_c5_write WRITE_SIZE $C5 --steps 4 --warmup 1 || exit 1
