#!/bin/bash
# A/B in separate processes, alternating: completion wait = poll-then-block (VF_SPIN_US=3000, default) vs block (0); then the text legs
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_spin_wait_ab.log
: > $L
for rep in 1 2 3; do
  for spin in 0 3000; do
    for rows in 1000000 1250000 10000000; do
      steps=300; [ $rows = 10000000 ] && steps=60
      VF_SPIN_US=$spin timeout -k 10 200 python3 bench.py --gpus 1 --rows $rows --steps $steps --warmup 20 --no-rerank --no-cpu-baseline --no-shard-legs --no-startup > gpurun_out/_ab.json 2>/dev/null || { echo fail; exit 1; }
      python3 - $rep $spin $rows <<'PY' >> $L
import json, sys
j = json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1]); r = j["roofline"]
print(f"rep {sys.argv[1]} VF_SPIN_US {sys.argv[2]:>4s} rows {sys.argv[3]:>8s}: {j['ms_per_step']:.4f} ms/step  p50 {j['p50_ms_per_step']}  interval frac {r['frac']}  host entry {j['host_entry']['ms_per_batch']}")
PY
    done
  done
done
cat $L
python3 - <<'PY'
import json, sys, os
sys.argv = ["bench.py"]
sys.path.insert(0, os.getcwd())
import bench
args = bench.parse()
print(json.dumps(bench.texts_legs(args), indent=None)[:3000])
PY
