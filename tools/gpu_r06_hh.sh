#!/bin/bash
# after the movrel candidate loop: cycle accounting once more, then fuzz soaks and the 40-run stress
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r06_scan2r_cycle_accounting_final.log
: > $L
for spec in "10000000 768 f16" "1250000 768 f16" "10000000 768 fp8"; do
  set -- $spec
  VF_LIB_PATH=$PWD/veritasfi_amd/lib/libvf_test.so VF_DBG_EXTRA=4096 timeout -k 10 200 python3 tools/stamps_scan2r.py $1 $2 $3 >> $L 2>gpurun_out/_st.err || { tail -5 gpurun_out/_st.err; exit 1; }
done
grep "debug+\| % of\|clock" $L
VF_FUZZ_SCAN2R=1 timeout -k 10 400 python3 tools/fuzz_search.py --seconds 200 --seed 101 > gpurun_out/r06_fuzz_scan2r_seed101.log 2>&1 || { tail -20 gpurun_out/r06_fuzz_scan2r_seed101.log; exit 1; }
tail -1 gpurun_out/r06_fuzz_scan2r_seed101.log
timeout -k 10 400 python3 tools/fuzz_search.py --seconds 240 --seed 102 > gpurun_out/r06_fuzz_seed102.log 2>&1 || { tail -20 gpurun_out/r06_fuzz_seed102.log; exit 1; }
tail -1 gpurun_out/r06_fuzz_seed102.log
timeout -k 10 300 python3 tools/fuzz_search.py --seconds 100 --seed 103 --repeat 4 > gpurun_out/r06_fuzz_seed103_repeat4.log 2>&1 || { tail -20 gpurun_out/r06_fuzz_seed103_repeat4.log; exit 1; }
tail -1 gpurun_out/r06_fuzz_seed103_repeat4.log
timeout -k 10 300 python3 tools/stress_repeat.py --runs 40 > gpurun_out/r06_stress_repeat_40_final.log 2>&1 || { tail -20 gpurun_out/r06_stress_repeat_40_final.log; exit 1; }
grep -c '"failures": 0' gpurun_out/r06_stress_repeat_40_final.log; grep -v '"failures": 0' gpurun_out/r06_stress_repeat_40_final.log | head -3
