#!/bin/bash
# clock and socket power while one product runs in a loop, on random and on all-zero operands; MFMA-only rates on smooth / random operands
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r04_power.log
: > $L
echo "== registers only" >> $L
[ -x veritasfi_amd/lib/mfma_rate16 ] || hipcc --offload-arch=gfx950 -O3 -w -o veritasfi_amd/lib/mfma_rate16 tools/ubench/mfma_rate16.hip || exit 1
timeout -k 10 120 veritasfi_amd/lib/mfma_rate16 >> $L 2>&1 || { tail $L; exit 1; }
for data in random zeros; do
  for kind in 10 0v; do
    echo "== loop: kind $kind data $data" >> $L
    timeout -k 10 60 python tools/gemm_loop.py --kind $kind --data $data --seconds 8 >> $L 2>&1 &
    pid=$!
    sleep 5
    for i in 1 2 3; do rocm-smi -c -P 2>/dev/null | grep -E "sclk|Power|power" >> $L; sleep 0.7; done
    wait $pid || { tail $L; exit 1; }
  done
done
echo "== idle" >> $L
rocm-smi -c -P 2>/dev/null | grep -E "sclk|Power|power" >> $L
cat $L
