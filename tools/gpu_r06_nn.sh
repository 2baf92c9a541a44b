#!/bin/bash
# the final tree once more: whole GPU suite, the driver's command, then more fuzz seeds
set -o pipefail
bash tools/gpu_r06_record.sh a || exit 1
VF_FUZZ_SCAN2R=1 timeout -k 10 400 python3 tools/fuzz_search.py --seconds 200 --seed 121 > gpurun_out/r06_fuzz_scan2r_seed121.log 2>&1 || { tail -20 gpurun_out/r06_fuzz_scan2r_seed121.log; exit 1; }
tail -1 gpurun_out/r06_fuzz_scan2r_seed121.log | cut -c1-120
timeout -k 10 400 python3 tools/fuzz_search.py --seconds 200 --seed 122 --max-work 6e10 > gpurun_out/r06_fuzz_seed122.log 2>&1 || { tail -20 gpurun_out/r06_fuzz_seed122.log; exit 1; }
tail -1 gpurun_out/r06_fuzz_seed122.log | cut -c1-120
