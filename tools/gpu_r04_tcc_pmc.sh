#!/bin/bash
# L2 hit rate of the large products in the 100-pair forward (kernel-trace + pmc only)
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/tccpmc
cd /tmp && export TMPDIR=/tmp
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_READ_sum TCC_WRITE_sum"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' _)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $REPO/gpurun_out/tccpmc -o $tag -- python3 $REPO/tools/bench_rerank.py --shape xlmr-base > $REPO/gpurun_out/tccpmc/run_$tag.log 2>&1 || { tail -3 $REPO/gpurun_out/tccpmc/run_$tag.log; echo "(set $set failed)"; }
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/tccpmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc, key=lambda k: -len(acc[k]))[:8]:
    if "gemm" not in k and "attention" not in k and "layernorm" not in k: continue
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"   {c:28s} mean per launch {sum(v)/len(v):16.1f}  ({len(v)} launches)")
PY
find gpurun_out/tccpmc -name "*.csv" -delete
