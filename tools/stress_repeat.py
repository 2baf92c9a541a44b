#!/usr/bin/env python3
"""Repeat-run stress: the SAME search many times over (index built once), every run compared bit for bit with the oracle -- finds
timing-dependent faults a single run of a fuzz case can miss (round 4: the fuzz had found an intermittent wrong score in the narrow
fp8-MFMA scan variant; this loop showed 12 of 40 runs losing one row, and that none of the shipped kernels do).
    python tools/stress_repeat.py [--runs 40]"""
import argparse, importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

CASES = [
    # (what, case, options)
    ("narrow fp8 rows, 2 queries, k = 2048 (k_scan)", dict(dtype="fp8", d=1536, nq=2, n=54938, k=2048, data="normal", seed=982935475), {}),
    ("narrow fp8 rows, same, k_scan2 with converted rows", dict(dtype="fp8", d=1536, nq=2, n=54938, k=2048, data="normal", seed=982935475), {"scan_impl": 3}),
    ("narrow fp16 rows, 64 queries (k_scan2)", dict(dtype="f16", d=768, nq=64, n=120000, k=100, data="normal", seed=11), {}),
    ("narrow fp16 rows, 3 queries, clustered data", dict(dtype="f16", d=1024, nq=3, n=200000, k=1000, data="clusters", seed=12), {}),
    ("wide fp8 rows, 200 queries (k_scan_wide8)", dict(dtype="fp8", d=1024, nq=200, n=60000, k=1000, data="normal", seed=13), {}),
    ("wide fp8 rows, 300 queries, duplicates (k_scan_wide8)", dict(dtype="fp8", d=768, nq=300, n=40000, k=100, data="dupes", seed=14), {}),
    ("wide fp8 rows, 1024 queries, k = 1000: the configs[4] call shape (k_scan_wide8)", dict(dtype="fp8", d=1024, nq=1024, n=150000, k=1000, data="normal", seed=17), {}),
    ("wide fp8 rows on the fp16 instruction (k_scan_wide)", dict(dtype="fp8", d=1024, nq=200, n=60000, k=1000, data="normal", seed=13), {"wide_mfma": 0}),
    ("wide fp16 rows, 96 queries (k_scan_wide)", dict(dtype="f16", d=768, nq=96, n=80000, k=100, data="normal", seed=15), {}),
    ("sharded handle, 3 row blocks, fp16", dict(dtype="f16", d=768, nq=64, n=150000, k=100, data="normal", seed=16, shards=3), {}),
    ("narrow fp16 rows, 64 queries: k_scan2r for the scan and the sample pass (round 6)", dict(dtype="f16", d=768, nq=64, n=200000, k=100, data="normal", seed=18), {"scan_impl": 5, "sample_impl": 1}),
    ("narrow fp16 rows, 20 queries, clustered data: k_scan2r<1>", dict(dtype="f16", d=768, nq=20, n=150000, k=1000, data="clusters", seed=19), {"scan_impl": 5, "sample_impl": 1}),
    ("narrow fp32 rows (fp16 scan copy), 64 queries, duplicates: k_scan2r", dict(dtype="f32", d=768, nq=64, n=100000, k=100, data="dupes", seed=20), {"scan_impl": 5, "sample_impl": 1}),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=40)
    a = ap.parse_args()
    spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tools", "fuzz_search.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    import veritasfi_amd as vf
    from oracle import canonical
    canonical.build()
    total_bad = 0
    for what, case, opts in CASES:
        codes, rows, q = fz.make_data(case)
        oi, os_ = canonical.search(rows, q, case["k"])
        dev_ids = [0] * case.get("shards", 1) if case.get("shards", 1) > 1 else None
        ix = vf.DenseIndex.from_e4m3(codes, device_ids=dev_ids) if codes is not None else vf.DenseIndex(rows, device_ids=dev_ids)
        for key, val in opts.items():
            ix.set_option(key, val)
        bad = 0
        for _ in range(a.runs):
            ids, sc = ix.search(q, case["k"])
            if not (np.array_equal(oi, ids) and np.array_equal(os_.view(np.uint32), sc.view(np.uint32))):
                bad += 1
        st = ix.stats()
        ix.close()
        total_bad += bad
        print(json.dumps({"what": what, "runs": a.runs, "failures": bad, "scan_kernel": st.get("scan_kernel"), "path": st.get("path"),
                          "exact_reruns": st.get("exact_reruns")}), flush=True)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
