#!/bin/bash
# round 5: matrix-pipe utilisation by counter (north_star: "rocprof-reported ... MFMA utilisation against chip peak"): SQ_VALU_MFMA_BUSY_CYCLES over
# 4 x SQ_BUSY_CU_CYCLES per kernel, for the 100-pair re-rank forward, configs[4]'s wide scan and the headline scan.  One --pmc pass each (kernel-trace + pmc only).
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/mfmapmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="--kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES --output-format csv"
timeout -k 10 300 rocprofv3 $P -d $OUT -o fwd -- python3 $REPO/tools/bench_rerank.py --shape xlmr-base --pairs 100 --iters 4 > $OUT/run_fwd.log 2>&1 || { tail -3 $OUT/run_fwd.log; exit 1; }
timeout -k 10 300 rocprofv3 $P -d $OUT -o c5 -- python3 $REPO/bench.py --rows 10000000 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --steps 4 --warmup 1 > $OUT/run_c5.log 2>&1 || { tail -3 $OUT/run_c5.log; exit 1; }
timeout -k 10 300 rocprofv3 $P -d $OUT -o head -- python3 $REPO/bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-rerank --no-shard-legs > $OUT/run_head.log 2>&1 || { tail -3 $OUT/run_head.log; exit 1; }
cd $REPO
python3 - <<'PY' | tee gpurun_out/r05_pmc_mfma_busy.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/mfmapmc/**/*counter_collection.csv", recursive=True):
    tag = "fwd" if "fwd" in f.split("/")[-1] else "c5" if "c5" in f.split("/")[-1] else "head"
    for r in csv.DictReader(open(f)):
        acc[(tag, r["Kernel_Name"][:60])][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES), mean over the launches of one rocprofv3 --pmc pass")
for (tag, k), c in sorted(acc.items(), key=lambda kv: (kv[0][0], -sum(kv[1].get("SQ_BUSY_CU_CYCLES", [0])))):
    if not any(s in k for s in ("gemm", "attention", "k_scan")): continue
    m, b = c.get("SQ_VALU_MFMA_BUSY_CYCLES", []), c.get("SQ_BUSY_CU_CYCLES", [])
    if not m or not b or sum(b) == 0: continue
    print(f"{tag:5s} {k:62s} launches {len(b):4d}  MFMA busy {sum(m) / (4 * sum(b)):.3f}")
PY
find gpurun_out/mfmapmc -name "*.csv" -delete
