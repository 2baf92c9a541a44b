#!/bin/bash
# after the new default widths: the whole GPU suite, then fuzz soaks (k_scan2r shapes incl. 1024 / 512 / 384; general)
set -o pipefail
mkdir -p gpurun_out
t0=$(date +%s)
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r06_cc_suite.log 2>&1 || { tail -30 gpurun_out/r06_cc_suite.log; exit 1; }
echo "suite wall $(( $(date +%s) - t0 )) s: $(tail -1 gpurun_out/r06_cc_suite.log)"
VF_FUZZ_SCAN2R=1 timeout -k 10 400 python3 tools/fuzz_search.py --seconds 240 --seed 91 > gpurun_out/r06_fuzz_scan2r_seed91.log 2>&1 || { tail -20 gpurun_out/r06_fuzz_scan2r_seed91.log; exit 1; }
tail -1 gpurun_out/r06_fuzz_scan2r_seed91.log
timeout -k 10 300 python3 tools/fuzz_search.py --seconds 120 --seed 92 > gpurun_out/r06_fuzz_seed92.log 2>&1 || { tail -20 gpurun_out/r06_fuzz_seed92.log; exit 1; }
tail -1 gpurun_out/r06_fuzz_seed92.log
