#!/bin/bash
# round 4: whole GPU suite + the chain against the vendor library + the driver's bench command
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/decoder_errors.jsonl
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -15 gpurun_out/pytest_gpu.log
[ $rc -ge 124 ] && exit $rc
L=gpurun_out/r04_gemm_chain_vendor_vs_repo.log
: > $L
timeout -k 10 200 python tools/bench_gemm_chain.py >> $L 2>&1
timeout -k 10 200 python tools/bench_gemm_chain.py --forward-epilogues >> $L 2>&1
timeout -k 10 200 python tools/bench_gemm_chain.py --hidden 1024 --ffn 4096 --rows 12800,25600,51200 >> $L 2>&1
grep "^{" $L | cut -c1-330
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver.log 2> gpurun_out/bench_driver.err; rc2=$?
python - <<'PY'
import json
l=[x for x in open('gpurun_out/bench_driver.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1])
    print({k:d[k] for k in ('value','ms_per_step','p50_ms_per_step','rerank_p50_ms')})
    for k in ('rerank','rerank_large','rerank_llm','embed'):
        v=d.get(k) or {}
        print(k, {x:v.get(x) for x in ('p50_ms','tflops','frac','ms_per_batch','chunks_per_s')})
    print(d['c4']['p50_ms'])
PY
[ $rc -ne 0 ] && exit $rc
exit $rc2
