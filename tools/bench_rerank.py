#!/usr/bin/env python3
"""Re-rank latency: score `pairs` (query, passage) pairs of `tokens` tokens with a random-weight
cross-encoder of the given shape (no checkpoints offline).  Prints p50 / min ms and achieved TFLOP/s."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import veritasfi_amd as vf

SHAPES = {"xlmr-base": (768, 12, 12, 3072, 250002), "xlmr-large": (1024, 24, 16, 4096, 250002),
          "bert-base": (768, 12, 12, 3072, 30522)}

def random_encoder(shape, head, seed=0, vocab=None):
    """vocab: override the embedding-table size (it does not enter the forward's cost; a small table keeps the host-side
    generation of random weights short)."""
    H, L, heads, F, V = SHAPES[shape]
    V = vocab or V
    cfg = dict(vocab=V, hidden=H, layers=L, heads=heads, ffn=F, max_pos=514, type_vocab=1,
               roberta_pad_idx=1 if shape.startswith("xlmr") else -1, pooling=0, normalize=0 if head else 1, head=head,
               ln_eps=1e-5)
    from veritasfi_amd import _ffi
    import ctypes
    c = _ffi.EncoderConfig(**cfg); n16 = _ffi.c_i64(0); n32 = _ffi.c_i64(0)
    _ffi.lib().vf_encoder_weight_sizes(ctypes.byref(c), ctypes.byref(n16), ctypes.byref(n32))
    rng = np.random.default_rng(seed)
    w16 = (rng.standard_normal(n16.value, dtype=np.float32) * 0.02).astype(np.float16)
    w32 = rng.standard_normal(n32.value, dtype=np.float32) * 0.02
    # LayerNorm gammas ~ 1: they are scattered through w32; 1 + noise everywhere is fine for timing
    w32 += 0.5
    return vf.HipEncoder(cfg, w16, w32), cfg

def flops(cfg, b, t):
    H, F, L = cfg["hidden"], cfg["ffn"], cfg["layers"]
    return b * L * (2 * t * (4 * H * H + 2 * H * F) + 4 * t * t * H)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="xlmr-base")
    ap.add_argument("--pairs", type=int, default=100)
    ap.add_argument("--tokens", type=int, default=512)
    ap.add_argument("--iters", type=int, default=15)
    ap.add_argument("--head", type=int, default=1)
    a = ap.parse_args()
    enc, cfg = random_encoder(a.shape, a.head)
    rng = np.random.default_rng(99)
    ids = rng.integers(5, cfg["vocab"], size=(a.pairs, a.tokens)).astype(np.int32)
    mask = np.ones_like(ids)
    enc.forward(ids, mask)
    ts = []
    for _ in range(a.iters):
        t0 = time.perf_counter(); enc.forward(ids, mask); ts.append((time.perf_counter() - t0) * 1e3)
    p50 = float(np.median(ts)); fl = flops(cfg, a.pairs, a.tokens)
    print(json.dumps({"shape": a.shape, "pairs": a.pairs, "tokens": a.tokens, "p50_ms": round(p50, 3),
                      "min_ms": round(min(ts), 3), "tflops_at_p50": round(fl / p50 / 1e9, 1),
                      "frac_of_2.5PF": round(fl / p50 / 1e9 / 2500, 4)}))
    enc.close()

if __name__ == "__main__":
    main()
