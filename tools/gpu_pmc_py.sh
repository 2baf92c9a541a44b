#!/bin/bash
# SQ counter pass over any python tool (kernel-trace + pmc only).  Usage: gpu_pmc_py.sh <tag> "<counters>" <script.py> [args...]
set -o pipefail
TAG=$1; CTRS=$2; SCRIPT=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $REPO/gpurun_out/$TAG -o $TAG -- python3 $REPO/$SCRIPT "$@" > $REPO/gpurun_out/$TAG/run.log 2>&1 || { tail -3 $REPO/gpurun_out/$TAG/run.log; exit 1; }
cd $REPO
f=$(find gpurun_out/$TAG -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot = {k: sum(sum(v) for v in c.values()) for k, c in acc.items()}
for k in sorted(acc, key=lambda k: -tot[k])[:4]:
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"   {c:28s} mean per launch {sum(v)/len(v):16.1f}  ({len(v)} launches)")
PY
cp $f gpurun_out/${TAG}_counters.csv
