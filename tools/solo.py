#!/usr/bin/env python3
"""Serialized searches (one batch at a time) for clean per-kernel durations under rocprofv3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import veritasfi_amd as vf
from bench import make_shard
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
corpus = make_shard(torch, 0, rows, 768, dev)
g = torch.Generator(device=dev); g.manual_seed(4321)
q = torch.randn((64, 768), generator=g, device=dev)
ix = vf.DenseIndex(corpus)
for _ in range(steps):
    ix.search_device(q, 100)
torch.cuda.synchronize()
print(ix.stats())
ix.close()
