#!/bin/bash
# split-K with XCD-local slices: parity, then the products of the 13/25/50/100-pair forwards with the cut on (mode 2) and off (mode 0)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_encoder.py -m gpu -q -x -p no:cacheprovider -k "splitk_tail or gemm_kernels" > gpurun_out/pytest_sk2.log 2>&1; rc=$?
tail -3 gpurun_out/pytest_sk2.log
if [ $rc -ne 0 ]; then grep -a "Error\|assert" gpurun_out/pytest_sk2.log | head -20; exit $rc; fi
grep -a "split-K tail" gpurun_out/pytest_sk2.log
: > gpurun_out/r03_sk2.log
for mode in 0 2; do
 for M in 6656 12800 25600 51200; do
  for hk in "768 3072" "1024 4096"; do
    set -- $hk; H=$1; F=$2
    VF_SK_MODE=$mode timeout -k 10 120 python3 tools/bench_gemm.py --check 0 --shapes ${M}x$((3*H))x$H --kind 7 --epi 0 >> gpurun_out/r03_sk2.log 2>&1 || exit 1
    VF_SK_MODE=$mode timeout -k 10 120 python3 tools/bench_gemm.py --check 0 --shapes ${M}x${H}x$H,${M}x${H}x$F --kind 7 --epi 2 >> gpurun_out/r03_sk2.log 2>&1 || exit 1
    VF_SK_MODE=$mode timeout -k 10 120 python3 tools/bench_gemm.py --check 0 --shapes ${M}x${F}x$H --kind 7 --epi 1 >> gpurun_out/r03_sk2.log 2>&1 || exit 1
  done
 done
done
python3 - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r03_sk2.log") if l.startswith("{")]
from collections import defaultdict
t = defaultdict(dict)
for r in rows:
    t[(r["shape"], r["epi"])][r.get("sk_mode")] = r["us"]
for (sh, epi), d in t.items():
    M, N, K = map(int, sh.split("x"))
    print(f"{sh:18s} epi {epi}  tiles {(M // 256) * (N // 256):5d}  unsplit {d.get(0)}  split {d.get(2)}")
PY
