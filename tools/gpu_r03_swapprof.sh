#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/r03_swap_layers.txt
for lib in "" libvf_noswap.so; do
  L=""; [ -n "$lib" ] && L="$R/veritasfi_amd/lib/$lib"
  for shape in xlmr-large; do
    rm -rf /tmp/prof_rr
    VF_LIB_PATH=$L timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rr -o out -- python3 $R/tools/bench_rerank.py --shape $shape --iters 4 > /tmp/rr.log 2>/dev/null
    echo "== lib=${lib:-shipped} $shape: $(tail -1 /tmp/rr.log | cut -c1-120)" >> $R/gpurun_out/r03_swap_layers.txt
    t=$(find /tmp/prof_rr -name "*kernel_trace.csv" | head -1)
    python3 $R/tools/trace_layer.py "$t" 24 >> $R/gpurun_out/r03_swap_layers.txt
  done
done
cat $R/gpurun_out/r03_swap_layers.txt | cut -c1-150
