#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import veritasfi_amd as vf
from veritasfi_amd import _ffi
from bench import make_shard
# usage: stamps_final.py [rows] [dim] [batch] [k] [f16|fp8]
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 64
k = int(sys.argv[4]) if len(sys.argv) > 4 else 100
dt = sys.argv[5] if len(sys.argv) > 5 else "f16"
dev = torch.device("cuda", 0)
corpus = make_shard(torch, 0, rows, dim, dev, dt)
g = torch.Generator(device=dev); g.manual_seed(4321)
q = torch.randn((nq, dim), generator=g, device=dev)
ix = vf.DenseIndex(corpus)
ix.set_option("debug", 256)
for _ in range(3):
    ix.search_device(q, k)
buf = np.zeros(nq * 8, dtype=np.uint64)
n = _ffi.lib().vf_index_debug_read(ix._h, 0, buf.ctypes.data, buf.size)
t = buf.reshape(-1, 8).astype(np.int64)
t0 = t[:, 0].min()
us = (t[:, :5] - t0) / 100.0
names = ["start", "ranked(approx)", "rescored", "ranked(canon)", "end"]
for i, nm in enumerate(names):
    x = us[:, i]
    print(f"{nm:16s} min {x.min():7.1f} p50 {np.median(x):7.1f} max {x.max():7.1f}")
d = np.diff(us, axis=1)
for i in range(4):
    print(f"phase {names[i]} -> {names[i+1]}: p50 {np.median(d[:, i]):7.1f} max {d[:, i].max():7.1f}")
print(ix.stats())
