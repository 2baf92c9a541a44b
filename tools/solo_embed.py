import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tools"))
from bench_rerank import random_encoder
shape = sys.argv[1] if len(sys.argv) > 1 else "bert-base"
enc, cfg = random_encoder(shape, head=0)
ids = np.random.default_rng(0).integers(5, cfg["vocab"], size=(1, 32)).astype(np.int32)
mask = np.ones_like(ids)
for _ in range(20): enc.forward(ids, mask)
