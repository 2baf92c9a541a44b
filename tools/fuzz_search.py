#!/usr/bin/env python3
"""Differential fuzz of vf_index_search against the CPU oracle: random (rows, dim, queries, k, dtype, data shape, options)
drawn to sit ON the dispatch boundaries (dense path <= 16384 rows, narrow / wide pass at 65 / 129 queries, 64-multiples
of dim for the fp8 instruction, k around the row count, sample rows 4 / 16), one handle or a handle over 2-5 row blocks
(all on device 0: the merge and the id offsets), every result compared bit for bit (ids and score bits) with
oracle.canonical.  Test infrastructure: the oracle is the checker, never the thing measured.

    python tools/fuzz_search.py --seconds 240 --seed 1 [--max-work 2e10]

Prints one line per case that FAILS (with the arguments to reproduce it) and a summary; exit code 1 on any failure."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def draw_case(rng, max_work):
    pick = lambda xs: xs[int(rng.integers(len(xs)))]
    dtype = pick(["f32", "f16", "f16", "fp8", "fp8"])
    d = pick([1, 7, 16, 33, 64, 100, 128, 200, 256, 384, 512, 768, 768, 1000, 1024, 1024, 1536, 2048, int(rng.integers(1, 2049))])
    nq = pick([1, 1, 2, 3, 8, 31, 63, 64, 65, 66, 100, 127, 128, 129, 130, 200, 256, 257, 300, int(rng.integers(1, 400))])
    n_edges = [1, 2, 15, 255, 256, 257, 1000, 16383, 16384, 16385, 16640, 20000, 32768, 40000, 65536, 70001, 131072, 200000, 300000]
    n = pick(n_edges + [int(np.exp(rng.uniform(0, np.log(400000))))] * 6)
    while float(n) * nq * d > max_work and n > 17000:
        n = max(17000, n // 2)
    while float(n) * nq * d > max_work and nq > 1:
        nq = max(1, nq // 2)
    k = pick([1, 2, 10, 100, 100, 128, 500, 1000, 1000, 2048, n, n + 3, max(1, n - 1), int(rng.integers(1, 2049))])
    k = max(1, min(k, 2048))
    data = pick(["normal", "normal", "dupes", "clusters", "scaled", "zeros", "sorted", "lowrank"])
    opts = {}
    if os.environ.get("VF_FUZZ_SCAN2R") == "1":   # soak of round 6's kernels: k_scan2r for the scan and for the sample pass -- fp16 rows of 384 / 512 / 768 / 1024 (1000 pads to it), e4m3 rows of 768 / 1024
        dtype = pick(["f16", "f16", "f32", "fp8", "fp8"])
        d = pick([768, 768, 1024, 512, 384, 1000]) if dtype != "fp8" else pick([768, 1024])
        opts["scan_impl"] = 5
        opts["sample_impl"] = 1
    elif rng.random() < 0.3:
        opts["scan_impl"] = pick([1, 2, 3, 4, 5])
        opts["sample_impl"] = pick([-1, 0, 1])
    if rng.random() < 0.2:
        opts["wide_mfma"] = pick([0, 1])
    if rng.random() < 0.2:
        opts["sample_rows"] = pick([1, 4, 16, 64])
    if rng.random() < 0.35:      # the other tuning knobs: every one of them is a speed setting, none may change a result
        knobs = {"force_path": [0, 1, 2], "margin": [0, 1, 8, 64, 2048], "cap": [0, 256, 1024, 16384], "waves": [0, 64, 1024, 8192],
                 "scan_g": [0, 1, 2, 3, 4], "refresh_every": [1, 4, 128, 256], "steal": [0, 1], "wide": [0, 1, 16, 65, 200],
                 "wide_sync": [-1, 0, 2, 8], "aux_cus": [0, 32, 64], "sample_grid": [0, 8, 64, 1024], "overlap_scans": [0, 1]}
        for name in rng.choice(sorted(knobs), size=int(rng.integers(1, 4)), replace=False):
            opts[str(name)] = int(pick(knobs[str(name)]))
    shards = int(pick([1, 1, 1, 2, 3, 5])) if n >= 8 else 1      # > 1: one handle over that many row blocks, all on device 0
    return dict(dtype=dtype, d=int(d), nq=int(nq), n=int(n), k=int(k), data=data, opts=opts, shards=shards, seed=int(rng.integers(1 << 31)))


def make_data(case):
    rng = np.random.default_rng(case["seed"])
    n, d, nq, data = case["n"], case["d"], case["nq"], case["data"]
    q = rng.standard_normal((nq, d)).astype(np.float32)
    c = rng.standard_normal((n, d)).astype(np.float32)
    if data == "dupes" and n >= 4:          # exact duplicate rows: ties broken by id
        src = rng.integers(0, n, max(1, n // 3))
        dst = rng.integers(0, n, src.size)
        c[dst] = c[src]
    elif data == "clusters":                # many rows near a few queries: dense candidate lists, stage flushes
        m = max(1, min(n, 600))
        who = rng.integers(0, nq, m)
        c[rng.choice(n, m, replace=False)] = q[who] + 0.05 * rng.standard_normal((m, d)).astype(np.float32)
    elif data == "scaled":                  # rows of very different norms (cosine must not care)
        c *= np.exp(rng.uniform(-6, 6, (n, 1))).astype(np.float32)
    elif data == "zeros":                   # zero rows and a zero query: score 0 by convention
        c[rng.integers(0, n, max(1, n // 50))] = 0.0
        q[rng.integers(0, nq)] = 0.0
    elif data == "lowrank":                 # everything in a 3-dimensional subspace: near-ties everywhere
        r = min(3, d)
        basis = rng.standard_normal((r, d)).astype(np.float32)
        c = (rng.standard_normal((n, r)).astype(np.float32) @ basis).astype(np.float32)
    if case["dtype"] == "fp8":
        import torch
        from oracle import ref_numpy as R
        s = 0.5 if data != "scaled" else 1.0
        x = torch.from_numpy(np.clip(c * s, -400, 400))
        codes = x.to(torch.float8_e4m3fn).view(torch.uint8).numpy().copy()
        rows = R.decode_e4m3(codes).astype(np.float16)     # exact: the decoded values ARE the corpus
        return codes, rows, q
    if case["dtype"] == "f16":
        c = np.clip(c, -60000, 60000).astype(np.float16)
    return None, c, q


def run_case(vf, oracle, case, repeat=1):
    codes, rows, q = make_data(case)
    if case["data"] == "sorted":            # score-sorted corpus (ascending for query 0): thresholds rise all the way through
        sims = oracle.cosine(q[:1], rows.astype(np.float32))[0]
        order = np.argsort(sims, kind="stable")
        rows = np.ascontiguousarray(rows[order])
        if codes is not None:
            codes = np.ascontiguousarray(codes[order])
    dev_ids = [0] * case.get("shards", 1) if case.get("shards", 1) > 1 else None
    ix = vf.DenseIndex.from_e4m3(codes, device_ids=dev_ids) if codes is not None else vf.DenseIndex(rows, device_ids=dev_ids)
    try:
        for key, val in case["opts"].items():
            try:
                ix.set_option(key, val)
            except RuntimeError:
                pass                        # an option the shape does not admit: the default stays
        try:
            first = ix.search(q, case["k"])
        except RuntimeError as e:            # a forced fused path the index cannot take is REFUSED (an error, not a crash): drop the knob
            if "forced fused path" not in str(e):
                raise
            ix.set_option("force_path", -1)
            first = ix.search(q, case["k"])
        runs = [first] + [ix.search(q, case["k"]) for _ in range(max(1, repeat) - 1)]   # repeated runs of one case: timing-dependent faults
        st = ix.stats()
    finally:
        ix.close()
    oi, os_ = oracle.search(rows, q, case["k"])
    bad_runs = [r for r, (i_, s_) in enumerate(runs) if not (np.array_equal(oi, i_) and np.array_equal(os_.view(np.uint32), s_.view(np.uint32)))]
    ok = not bad_runs
    ids, sc = runs[bad_runs[0]] if bad_runs else runs[0]
    why = None
    if not ok:
        badq = np.nonzero((oi != ids).any(axis=1) | (os_.view(np.uint32) != sc.view(np.uint32)).any(axis=1))[0]
        why = {"queries": badq[:6].tolist(), "n_bad": int(badq.size), "bad_runs": bad_runs, "of_runs": len(runs)}
        b = int(badq[0])
        j = int(np.nonzero((oi[b] != ids[b]) | (os_[b].view(np.uint32) != sc[b].view(np.uint32)))[0][0])
        why["first"] = {"rank": j, "got": [int(ids[b, j]), float(sc[b, j])], "want": [int(oi[b, j]), float(os_[b, j])]}
    return ok, st, why


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-work", type=float, default=2e10, help="cap on rows x queries x dim per case (the oracle's cost)")
    ap.add_argument("--case", default=None, help="JSON of one case to re-run")
    ap.add_argument("--repeat", type=int, default=1, help="searches per case on the same index (all compared with the oracle)")
    a = ap.parse_args()
    import veritasfi_amd as vf
    from veritasfi_amd import _ffi
    _ffi.lib()
    from oracle import canonical
    canonical.build()
    if a.case:
        case = json.loads(a.case)
        ok, st, why = run_case(vf, canonical, case, a.repeat)
        print("OK" if ok else "FAIL", json.dumps(case), st, why)
        return 0 if ok else 1
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    n_cases = n_fail = 0
    paths = {}
    while time.time() - t0 < a.seconds:
        case = draw_case(rng, a.max_work)
        try:
            ok, st, why = run_case(vf, canonical, case, a.repeat)
        except Exception as e:   # noqa: BLE001 -- a fuzz driver reports everything
            ok, st, why = False, {}, {"exception": repr(e)}
        n_cases += 1
        key = (st.get("path"), st.get("scan_kernel"), case["dtype"])
        if case.get("shards", 1) > 1:
            paths[("sharded handle",)] = paths.get(("sharded handle",), 0) + 1
        paths[key] = paths.get(key, 0) + 1
        if st.get("exact_reruns"):
            paths[("exact_reruns",)] = paths.get(("exact_reruns",), 0) + 1
        if not ok:
            n_fail += 1
            print("FAIL", json.dumps(case), why, st, flush=True)
        if n_cases % 25 == 0:
            print(f"... {n_cases} cases, {n_fail} failures, {time.time() - t0:.0f} s", flush=True)
    print(json.dumps({"cases": n_cases, "failures": n_fail, "seconds": round(time.time() - t0, 1), "seed": a.seed,
                      "by_path_kernel_dtype": {str(k): v for k, v in sorted(paths.items(), key=str)}}))
    return 1 if n_fail else 0


if __name__ == "__main__":
    sys.exit(main())
