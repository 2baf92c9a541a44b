#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r05_where.log
: > $L
for w in 4 8; do
  VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_stamps.so timeout -k 10 300 python tools/stamps_wide8_where.py 10000000 $w 2>&1 | grep -v amdgpu.ids | tee -a $L
done
echo "== padded LDS (one 4-wave workgroup per CU)" | tee -a $L
VF_W8_PAD_LDS=20000 timeout -k 10 300 python bench.py --rows 10000000 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --steps 6 --warmup 2 --opt wide8_waves=4 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('  launch', j['roofline']['avg_launch_ms'], 'ms/step', j['ms_per_step'])" | tee -a $L
