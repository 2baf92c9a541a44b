#!/bin/bash
# which kernels does the vendor library pick for these shapes?  (names encode macro-tile, wave layout, LDS staging, split)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_v
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v -o out -- python3 $R/tools/bench_gemm.py --iters 5 --kind 7 --shapes 6656x2304x768,6656x768x768,6656x768x3072,6656x3072x768,12800x768x3072,25600x2304x768,51200x2304x768,51200x768x768,51200x3072x768,51200x768x3072,51200x1024x4096,51200x4096x1024 > $R/gpurun_out/vendor_names_bench.log 2>&1
t=$(find /tmp/prof_v -name "*kernel_trace.csv" | head -1)
python3 - "$t" > $R/gpurun_out/r03_vendor_kernels.txt <<'PY'
import csv, sys
from collections import OrderedDict
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
seen = OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "Cijk" in n or "gemm" in n.lower():
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (n, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""))
        seen.setdefault(key, []).append(d)
for (n, g, w, lds, vg, ag), v in seen.items():
    print(f"{min(v):8.1f} us x{len(v):3d} grid {g} wg {w} lds {lds} vgpr {vg} agpr {ag}  {n[:400]}")
PY
cat $R/gpurun_out/r03_vendor_kernels.txt | cut -c1-600
