#!/usr/bin/env python3
"""Decoder-only forward latency (random weights of a named shape; no checkpoints offline): last-token embeddings of
`batch` sequences of `tokens` tokens.  Shapes: Qwen3-Embedding 0.6B / 4B (the reference's default embedder,
experiments/retriever/step3_mul.py:384)."""
import argparse, ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import veritasfi_amd as vf
from veritasfi_amd import _ffi

SHAPES = {  # hidden, layers, heads, kv_heads, head_dim, ffn, vocab
    "qwen3-0.6b": (1024, 28, 16, 8, 128, 3072, 151669),
    "qwen3-4b": (2560, 36, 32, 8, 128, 9728, 151669),
    "gemma-2b": (2048, 18, 8, 1, 256, 16384, 256000),
    "tiny": (256, 2, 4, 2, 64, 512, 1000),
}

def decoder_flops(shape, batch, tokens):
    H, L, NH, KV, DH, F, V = SHAPES[shape]
    QD, KD = NH * DH, KV * DH
    per_tok = 2 * (H * (QD + 2 * KD) + QD * H + 3 * H * F)
    return batch * L * (tokens * per_tok + 2 * tokens * tokens * QD)      # causal attention: half of 4 T^2 QD


def random_decoder(shape, score_token=None, vocab=None, pooling=2, normalize=1):
    """Random-weight HipDecoder of a named shape; the weights are generated on the GPU (5 GB of fp16 at the gemma-2b shape
    would take the host generator half a minute) and handed to the C ABI as a host blob.  vocab: override the embedding
    table's rows (it does not enter the forward's cost)."""
    import torch
    H, L, NH, KV, DH, F, V = SHAPES[shape]
    V = vocab or V
    gemma = shape.startswith("gemma")
    cfg = dict(vocab=V, hidden=H, layers=L, heads=NH, kv_heads=KV, head_dim=DH, ffn=F, rope_theta=1e6, rms_eps=1e-6,
               qk_norm=0 if gemma else 1, pooling=pooling, normalize=normalize, head=2 if score_token is not None else 0,
               act=1 if gemma else 0, norm_plus_one=1 if gemma else 0, embed_scale=float(H ** 0.5) if gemma else 1.0)
    c = _ffi.DecoderConfig(**cfg); n16 = _ffi.c_i64(0); n32 = _ffi.c_i64(0)
    _ffi.check(_ffi.lib().vf_decoder_weight_sizes(ctypes.byref(c), ctypes.byref(n16), ctypes.byref(n32)), "sizes")
    g = torch.Generator(device="cuda").manual_seed(0)
    w16 = torch.empty(n16.value, dtype=torch.float16)
    for i in range(0, n16.value, 1 << 28):
        m = min(1 << 28, n16.value - i)
        w16[i:i + m] = (torch.randn(m, generator=g, device="cuda", dtype=torch.float32) * 0.02).half().cpu()
    w32 = np.zeros(n32.value, np.float32) if gemma else np.ones(n32.value, np.float32)
    return vf.HipDecoder(cfg, w16.numpy(), w32), cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="qwen3-0.6b")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--tokens", type=int, default=512)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--ragged", type=int, default=0, help="lengths uniform in [ragged, tokens], LEFT-padded (the LLM re-ranker's input form)")
    a = ap.parse_args()
    H, L, NH, KV, DH, F, V = SHAPES[a.shape]
    cfg = dict(vocab=V, hidden=H, layers=L, heads=NH, kv_heads=KV, head_dim=DH, ffn=F, rope_theta=1e6, rms_eps=1e-6,
               qk_norm=0 if a.shape.startswith("gemma") else 1, pooling=2, normalize=1, head=0,
               act=1 if a.shape.startswith("gemma") else 0, norm_plus_one=1 if a.shape.startswith("gemma") else 0,
               embed_scale=float(H ** 0.5) if a.shape.startswith("gemma") else 1.0)
    c = _ffi.DecoderConfig(**cfg); n16 = _ffi.c_i64(0); n32 = _ffi.c_i64(0)
    _ffi.check(_ffi.lib().vf_decoder_weight_sizes(ctypes.byref(c), ctypes.byref(n16), ctypes.byref(n32)), "sizes")
    rng = np.random.default_rng(0)
    w16 = np.empty(n16.value, np.float16)
    for i in range(0, n16.value, 1 << 26):          # chunked: the 4B shape is 8 GB of fp16
        m = min(1 << 26, n16.value - i)
        w16[i:i + m] = (rng.standard_normal(m, dtype=np.float32) * 0.02).astype(np.float16)
    w32 = np.zeros(n32.value, np.float32) if a.shape.startswith("gemma") else np.ones(n32.value, np.float32)
    dec = vf.HipDecoder(cfg, w16, w32)
    del w16
    ids = rng.integers(5, V, size=(a.batch, a.tokens)).astype(np.int32)
    mask = np.ones_like(ids)
    if a.ragged:
        lens = rng.integers(a.ragged, a.tokens + 1, size=a.batch)
        lens[0] = a.tokens
        mask = (np.arange(a.tokens)[None, :] >= (a.tokens - lens)[:, None]).astype(np.int32)
    dec.forward(ids, mask)
    ts = []
    for _ in range(a.iters):
        t0 = time.perf_counter(); out = dec.forward(ids, mask); ts.append((time.perf_counter() - t0) * 1e3)
    QD, KD, T = NH * DH, KV * DH, a.tokens
    per_tok = 2 * (H * (QD + 2 * KD) + QD * H + 3 * H * F)
    flops = a.batch * L * (T * per_tok + 2 * T * T * QD)      # causal attention: half of 4 T^2 QD
    p50 = float(np.median(ts))
    print(json.dumps({"shape": a.shape, "batch": a.batch, "tokens": a.tokens, "ragged_from": a.ragged, "valid_tokens": int(mask.sum()),
                      "packed": os.environ.get("VF_NO_PACKED") is None, "p50_ms": round(p50, 2), "min_ms": round(min(ts), 2),
                      "tflops_at_p50": round(flops / p50 / 1e9, 1), "seq_per_s": round(a.batch / p50 * 1e3, 1),
                      "finite": bool(np.isfinite(out).all())}))
    dec.close()

if __name__ == "__main__":
    main()
