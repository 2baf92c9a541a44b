#!/bin/bash
# is the large product's main loop bound by the LDS array?  SQ counters over the 100-pair forward (kernel-trace + pmc only)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/ldspmc
cd /tmp && export TMPDIR=/tmp
for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT"; do
  tag=$(echo $set | cut -c1-14 | tr ' ' _)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $REPO/gpurun_out/ldspmc -o $tag -- python3 $REPO/tools/bench_rerank.py --shape xlmr-base > $REPO/gpurun_out/ldspmc/run_$tag.log 2>&1 || { tail -3 $REPO/gpurun_out/ldspmc/run_$tag.log; exit 1; }
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/ldspmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc, key=lambda k: -len(acc[k]))[:8]:
    if "gemm" not in k and "attention" not in k: continue
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"   {c:28s} mean per launch {sum(v)/len(v):16.1f}  ({len(v)} launches)")
PY
find gpurun_out/ldspmc -name "*.csv" -delete
