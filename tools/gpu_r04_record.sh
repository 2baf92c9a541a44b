#!/bin/bash
# round 4 record: kernel stats + PMC passes of the headline workload, configs[1] / the 8-GPU shard rehearsal / configs[4] lines,
# the reference harnesses' counterparts
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
echo "== kernel stats, the driver's command"; bash tools/gpu_prof_bench.sh r04_10m stats --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-rerank || exit 1
echo "== PMC, 10M x 768"; bash tools/gpu_prof_bench.sh r04_10m_fetch FETCH_SIZE --steps 6 --warmup 2 --no-cpu-baseline --no-rerank || exit 1
bash tools/gpu_prof_bench.sh r04_10m_write WRITE_SIZE --steps 6 --warmup 2 --no-cpu-baseline --no-rerank || exit 1
echo "== configs[1]: 1M x 768"; timeout -k 10 300 python bench.py --rows 1000000 --steps 200 --warmup 20 --no-rerank > gpurun_out/r04_bench_c2_1Mx768.log 2>&1 || exit 1
tail -1 gpurun_out/r04_bench_c2_1Mx768.log | cut -c1-400
echo "== the 8-GPU shard, one self-launched rank, exchange forced"
VF_BENCH_LAUNCH=1 VF_BENCH_FORCE_EXCHANGE=1 timeout -k 10 300 python bench.py --gpus 1 --rows 1250000 --steps 200 --warmup 20 --verify --no-rerank > gpurun_out/r04_bench_rehearsal_1250k_exchange.log 2>&1 || exit 1
tail -1 gpurun_out/r04_bench_rehearsal_1250k_exchange.log | cut -c1-400
C5="--rows 10000000 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank"
echo "== configs[4]"; timeout -k 10 400 python bench.py $C5 --steps 10 --warmup 2 > gpurun_out/r04_bench_c5.log 2>&1 || exit 1
tail -1 gpurun_out/r04_bench_c5.log | cut -c1-400
timeout -k 10 400 python bench.py $C5 --steps 10 --warmup 2 --opt wide_mfma=0 > gpurun_out/r04_bench_c5_f16mfma.log 2>&1 || exit 1
tail -1 gpurun_out/r04_bench_c5_f16mfma.log | cut -c1-300
timeout -k 10 400 python bench.py --rows 1250000 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --steps 20 --warmup 3 > gpurun_out/r04_bench_c5_shard_1250k.log 2>&1 || exit 1
tail -1 gpurun_out/r04_bench_c5_shard_1250k.log | cut -c1-300
echo "== configs[4]: kernel stats + PMC"; bash tools/gpu_prof_bench.sh r04_c5 stats $C5 --steps 10 --warmup 2 || exit 1
bash tools/gpu_prof_bench.sh r04_c5_fetch FETCH_SIZE $C5 --steps 6 --warmup 2 || exit 1
bash tools/gpu_prof_bench.sh r04_c5_write WRITE_SIZE $C5 --steps 6 --warmup 2 || exit 1
echo "== harness counterparts"
timeout -k 10 300 python tools/continuous_retrieval.py > gpurun_out/r04_continuous_retrieval.log 2>&1; tail -3 gpurun_out/r04_continuous_retrieval.log
timeout -k 10 300 python tools/rerank_stress.py > gpurun_out/r04_rerank_stress.log 2>&1; tail -3 gpurun_out/r04_rerank_stress.log
echo "== vision + clip text"
timeout -k 10 200 python tools/bench_vision.py > gpurun_out/r04_vision_bench.log 2>&1; tail -1 gpurun_out/r04_vision_bench.log
echo "== decoders"
timeout -k 10 300 python tools/bench_decoder.py > gpurun_out/r04_decoder_bench.log 2>&1; tail -3 gpurun_out/r04_decoder_bench.log
