#!/bin/bash
# round 5 soak: the differential fuzzers over the tree as shipped (new this round: relaxed split-K arrival, 8-phase tail between one and two rounds,
# half items in k_attention2, per-handle streams): encoder (small and --big), product dispatch, decoder, search
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
L=$R/gpurun_out/r05_fuzz_soak.log
: > $L
echo "== fuzz_gemm 150 s" >> $L;            timeout -k 10 220 python tools/fuzz_gemm.py --seconds 150 --seed 505 2>&1 | tail -4 >> $L || { tail -20 $L; exit 1; }
echo "== fuzz_encoder 120 s" >> $L;         timeout -k 10 200 python tools/fuzz_encoder.py --seconds 120 --seed 506 2>&1 | tail -4 >> $L || { tail -20 $L; exit 1; }
echo "== fuzz_encoder --big 150 s" >> $L;   timeout -k 10 260 python tools/fuzz_encoder.py --big --seconds 150 --seed 507 2>&1 | tail -4 >> $L || { tail -20 $L; exit 1; }
echo "== fuzz_decoder 90 s" >> $L;          timeout -k 10 180 python tools/fuzz_decoder.py --seconds 90 --seed 508 2>&1 | tail -4 >> $L || { tail -20 $L; exit 1; }
echo "== fuzz_search 150 s" >> $L;          timeout -k 10 260 python tools/fuzz_search.py --seconds 150 --seed 509 --repeat 2 2>&1 | tail -4 >> $L || { tail -20 $L; exit 1; }
cat $L | cut -c1-220
