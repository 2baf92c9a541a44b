#!/bin/bash
# rocprofv3 of a bench.py command: kernel stats (default) or one PMC counter.  Usage: gpu_prof_bench.sh <tag> <stats|FETCH_SIZE|WRITE_SIZE> [bench args...]
set -o pipefail
TAG=$1; MODE=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
if [ "$MODE" = "stats" ]; then
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/$TAG -o $TAG -- python3 $REPO/bench.py "$@" > $REPO/gpurun_out/$TAG/run.log 2>&1 || { tail -3 $REPO/gpurun_out/$TAG/run.log; exit 1; }
  cd $REPO
  f=$(find gpurun_out/$TAG -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/${TAG}_kernel_stats.csv
  head -9 $f | cut -c1-170
  find gpurun_out/$TAG -name "*.csv" ! -name "*stats*" -delete
else
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $MODE --output-format csv -d $REPO/gpurun_out/$TAG -o $TAG -- python3 $REPO/bench.py "$@" > $REPO/gpurun_out/$TAG/run.log 2>&1 || { tail -3 $REPO/gpurun_out/$TAG/run.log; exit 1; }
  cd $REPO
  f=$(find gpurun_out/$TAG -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$MODE" <<'PY'
import csv, sys, collections
f, ctr = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") == ctr:
        acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:6]:
    print(f"{ctr} {k}: launches={len(v)} mean={sum(v)/len(v):.1f} (KB as reported)")
PY
  cp $f gpurun_out/${TAG}_${MODE}.csv
fi
