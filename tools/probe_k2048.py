import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import veritasfi_amd as vf
from bench import make_shard
dev = torch.device("cuda", 0)
for rows in (5_000_000, 1_000_000):
    corpus = make_shard(torch, 0, rows, 768, dev)
    ix = vf.DenseIndex(corpus)
    g = torch.Generator(device=dev); g.manual_seed(1)
    for nq in (1, 4):
        q = torch.randn((nq, 768), generator=g, device=dev)
        for k in (100, 1000, 2048):
            ix.search_device(q, k); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); ix.search_device(q, k); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            st = ix.stats()
            print(rows, "nq", nq, "k", k, f"{np.median(ts)*1e3:.2f} ms", {kk: st[kk] for kk in ("path", "candidates", "max_candidates", "uncertified", "overflowed", "exact_reruns", "scan_kernel")}, flush=True)
    ix.close(); del corpus
