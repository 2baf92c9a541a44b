import os, sys, time, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, "tests"); sys.path.insert(0, os.getcwd())
    import numpy as np
    import tokenizers_synth as TS
    from veritasfi_amd.host_tokenize import BatchTokenizer
    rng = np.random.default_rng(0); W = TS.WORDS
    sent = lambda n: " ".join(W[i] for i in rng.integers(0, len(W), n))
    xt = TS.xlmr_tokenizer(); bt = BatchTokenizer(xt, 512)
    q = [sent(16) for _ in range(100)]; d = [sent(520) for _ in range(100)]
    bt.encode(q, d); ts = []
    for _ in range(8):
        t0 = time.perf_counter(); bt.encode(q, d); ts.append(time.perf_counter() - t0)
    ts2 = []
    for _ in range(8):
        t0 = time.perf_counter(); bt.encode(q[:50], d[:50]); ts2.append(time.perf_counter() - t0)
    print(sys.argv[1], "cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), f"100 pairs {np.median(ts)*1e3:.2f} ms, 50 pairs {np.median(ts2)*1e3:.2f} ms", flush=True)
else:
    for n in ("default", "4", "8", "16", "32", "64"):
        env = dict(os.environ)
        if n != "default": env["RAYON_NUM_THREADS"] = n
        subprocess.run([sys.executable, __file__, n], env=env)
