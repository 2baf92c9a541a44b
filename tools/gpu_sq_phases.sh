#!/bin/bash
# per-phase kernel times of the persistent one-query forward (run phase by phase).  Usage: gpu_sq_phases.sh [shape]
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/gpurun_out/sqph
cd /tmp && export TMPDIR=/tmp
VF_SQ_PHASES=1 VF_NO_GRAPH=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/sqph -o sqph -- python3 $REPO/tools/solo_embed.py ${1:-bert-base} > $REPO/gpurun_out/sqph/run.log 2>&1 || { tail -3 $REPO/gpurun_out/sqph/run.log; exit 1; }
cd $REPO
f=$(find gpurun_out/sqph -name "*kernel_trace.csv" | head -1)
python3 tools/sq_phase_times.py $f
rm -f $f
