#!/bin/bash
set -o pipefail
python3 - <<'PY'
import json, sys, os
sys.argv = ["bench.py"]
sys.path.insert(0, os.getcwd())
import bench
args = bench.parse()
t = bench.texts_legs(args)
e, r = t["embed_texts"], t["rerank_texts"]
print({k: e[k] for k in ("texts_per_s", "texts_per_s_one_call", "pre_tokenised_chunks_per_s", "ratio_to_pre_tokenised", "serial_loop_texts_per_s", "tokenize_100_ms", "bit_equal_to_serial_loop")})
print({k: r[k] for k in ("p50_ms", "pre_tokenised_p50_ms", "serial_loop_p50_ms", "tokenize_pairs_ms", "bit_equal_to_serial_loop")})
PY
