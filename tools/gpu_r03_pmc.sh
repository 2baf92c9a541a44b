#!/bin/bash
# HBM traffic of the main scan (k_scan2) from PMC counters: separate FETCH_SIZE / WRITE_SIZE passes with --kernel-trace only
# (MI355X_MICROARCH.md: FETCH_SIZE x 1024 x 2 on gfx950 for wide coalesced streaming reads + WRITE_SIZE x 1024)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc_$ctr -o out -- python3 $R/bench.py --rows ${ROWS:-10000000} --steps 6 --warmup 2 --no-cpu-baseline --no-rerank > $R/gpurun_out/pmc_$ctr.bench.log 2>&1 || { tail -3 $R/gpurun_out/pmc_$ctr.bench.log; exit 1; }
  f=$(find /tmp/pmc_$ctr -name "*counter_collection.csv" | head -1)
  cp "$f" $R/gpurun_out/r03_pmc_${ctr}_${TAG:-10Mx768}.csv
done
python3 - $R <<'PY'
import csv, json, sys, collections
R = sys.argv[1]
import os
tag = os.environ.get("TAG", "10Mx768")
rows = int(os.environ.get("ROWS", "10000000"))
out = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{R}/gpurun_out/r03_pmc_{ctr}_{tag}.csv")):
        if r.get("Counter_Name") == ctr:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    name = max((k for k in acc if "k_scan" in k and "Li0E" not in k.split("k_scan")[1][:12]), key=lambda k: sum(acc[k]))
    v = acc[name]
    # the main launches are the large ones (the sample pass shares the k_scan name only for the register kernel)
    big = [x for x in v if x > 0.5 * max(v)]
    out[ctr] = (name, sum(big) / len(big), len(big))
line = [json.loads(l) for l in open(f"{R}/gpurun_out/pmc_FETCH_SIZE.bench.log") if l.startswith("{")][0]
alg = line["roofline"]["bytes_per_launch"]
hbm = out["FETCH_SIZE"][1] * 1024 * 2 + out["WRITE_SIZE"][1] * 1024
rec = {"kernel": out["FETCH_SIZE"][0][:60], "workload": {"rows": rows, "dim": 768, "batch": 64, "k": 100, "n_gpus": 1},
       "FETCH_SIZE_mean_KB": out["FETCH_SIZE"][1], "WRITE_SIZE_mean_KB": out["WRITE_SIZE"][1], "launches": out["FETCH_SIZE"][2],
       "correction": "hbm_bytes = FETCH_SIZE*1024*2 (gfx950 reports half of a wide coalesced streaming read, LDS-DMA loads included) + WRITE_SIZE*1024; separate --pmc passes with --kernel-trace only",
       "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "ratio": hbm / alg,
       "measured": "round 3 (tools/gpu_r03_pmc.sh, profiles/r03_pmc_*)"}
json.dump(rec, open(f"{R}/gpurun_out/pmc_traffic_scan2_{tag}.json", "w"), indent=1)
print(json.dumps(rec, indent=1))
PY
