#!/bin/bash
# (build the timing libraries first, here, before gpurun -- they travel with the snapshot:
#   for cfg in noepi:-DVF_WIDE_EXP=8 noepi_nobar:-DVF_WIDE_EXP=12 noepi_noq:-DVF_WIDE_EXP=9 noepi_noops:-DVF_WIDE_EXP=11 noepi_nosib:-DVF_WIDE_EXP=24 nocand:-DVF_WIDE_NOCAND; do
#     VF_BUILD_FLAGS="${cfg#*:}" VF_BUILD_LIB=libvf_w_${cfg%%:*}.so VF_BUILD_TAG=_w_${cfg%%:*} python -m veritasfi_amd.build; done )
# where does k_scan_wide's launch go?  timing builds (results invalid) of the configs[4] shard (1.25M x 1024 e4m3, 1024 queries, k = 1000)
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r03_wide_ablate.log
for tag in shipped noepi noepi_nobar noepi_noq noepi_noops noepi_nosib; do
  L=""; nv=""
  if [ $tag != shipped ]; then L="$PWD/veritasfi_amd/lib/libvf_w_$tag.so"; nv="--no-verify"; fi
  echo -n "$tag: " >> gpurun_out/r03_wide_ablate.log
  VF_LIB_PATH=$L timeout -k 10 300 python3 bench.py --rows 1250000 --dim 1024 --batch 1024 --k 1000 --corpus-dtype fp8 --no-cpu-baseline --no-rerank --no-llm --no-c4 --steps 30 --warmup 3 --no-verify 2>/dev/null | grep -a "^{" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('launch ms', r.get('avg_launch_ms'), 'TFLOP/s', r.get('achieved'), 'cand/query', d['search_stats']['candidates_per_query'])
" >> gpurun_out/r03_wide_ablate.log 2>&1 || echo failed >> gpurun_out/r03_wide_ablate.log
done
cat gpurun_out/r03_wide_ablate.log
