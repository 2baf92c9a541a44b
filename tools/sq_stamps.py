#!/usr/bin/env python3
"""In-kernel phase clocks of the persistent one-query forward (k_sq_forward): per phase kind, the time from the barrier's
release to the slowest workgroup's end (body) and from there to the next phase's first begin (barrier).  VF_NO_GRAPH=1."""
import ctypes, os, sys
os.environ["VF_NO_GRAPH"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from veritasfi_amd import _ffi
from bench_rerank import random_encoder

shape = sys.argv[1] if len(sys.argv) > 1 else "bert-base"
t = int(sys.argv[2]) if len(sys.argv) > 2 else 32
L = _ffi.lib()
L.vf_debug_sq_stamps.argtypes = [ctypes.c_void_p]
L.vf_debug_sq_mode(1)
enc, cfg = random_encoder(shape, head=0)
ids = np.random.default_rng(0).integers(5, cfg["vocab"], size=(1, t)).astype(np.int32)
mask = np.ones_like(ids)
for _ in range(5): enc.forward(ids, mask)
nph, grid = 4 * cfg["layers"], 256
buf = torch.zeros(nph * grid * 4, dtype=torch.int64, device="cuda:0")
L.vf_debug_sq_stamps(buf.data_ptr())
for _ in range(3): enc.forward(ids, mask)
L.vf_debug_sq_stamps(None)
st = buf.cpu().numpy().reshape(nph, grid, 4).astype(np.float64) * 10.0     # ns
names = ["P1 ln+qkv+attention", "P2 o-proj", "P3 ln+ffn-up", "P4 ffn-down"]
body = [[] for _ in range(4)]; bar = [[] for _ in range(4)]; first = [[] for _ in range(4)]
for ph in range(nph):
    act = st[ph][:, 0] > 0
    b0, e1 = st[ph][act, 0].min(), st[ph][act, 1].max()
    body[ph % 4].append(e1 - b0)
    first[ph % 4].append(st[ph][act, 1].min() - b0)
    if ph + 1 < nph:
        nxt = st[ph + 1][:, 0]; nxt = nxt[nxt > 0]
        bar[ph % 4].append(nxt.min() - e1)
for k in range(4):
    print(names[k], "body(slowest) ns", round(float(np.median(body[k]))), " fastest wg", round(float(np.median(first[k]))),
          " barrier after it ns", round(float(np.median(bar[k]))) if bar[k] else None)
for k in (0, 1, 2, 3):
    d = [(st[ph][0, 2] - st[ph][0, 0], st[ph][0, 3] - st[ph][0, 0], st[ph][0, 1] - st[ph][0, 0]) for ph in range(k, nph, 4)]
    print(names[k], "workgroup 0: begin -> mfma done, -> reduced, -> end (ns):", np.median(np.array(d), axis=0))
print("total ns", st[st > 0].max() - st[st > 0].min())
enc.close()
