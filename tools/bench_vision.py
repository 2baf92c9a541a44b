#!/usr/bin/env python3
"""Throughput of the vision tower (vf_vit_*) at the geometry BASELINE configs[3] names: CLIP ViT-L/14 (24 layers, 1024 wide,
257 tokens per 224 x 224 image, 768-d projection), random-init weights (no checkpoints offline), seeded normal pixels.
Host buffers in and out (PCIe-inclusive), like the text embedder's embed loop."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

SHAPES = {"vit-l-14": dict(image=224, patch=14, channels=3, hidden=1024, layers=24, heads=16, ffn=4096, proj_dim=768),
          "vit-b-16": dict(image=224, patch=16, channels=3, hidden=768, layers=12, heads=12, ffn=3072, proj_dim=512)}


def random_vit(shape, seed=0):
    from veritasfi_amd import _ffi
    from veritasfi_amd.vision import HipVisionEncoder
    import ctypes
    cfg = dict(SHAPES[shape], act=1, normalize=1, ln_eps=1e-5)
    c = _ffi.VitConfig(**cfg)
    n16, n32 = _ffi.c_i64(0), _ffi.c_i64(0)
    _ffi.check(_ffi.lib().vf_vit_weight_sizes(ctypes.byref(c), ctypes.byref(n16), ctypes.byref(n32)))
    rng = np.random.default_rng(seed)
    w16 = (rng.standard_normal(n16.value, dtype=np.float32) * 0.03).astype(np.float16)
    w32 = np.zeros(n32.value, np.float32)
    H, F, L = cfg["hidden"], cfg["ffn"], cfg["layers"]
    per = 2 * H + 3 * H + H + 2 * H + F + H
    w32[:H] = 1.0
    for l in range(L):
        o = 2 * H + l * per
        w32[o:o + H] = 1.0
        w32[o + 6 * H:o + 7 * H] = 1.0
    w32[2 * H + L * per:2 * H + L * per + H] = 1.0
    return HipVisionEncoder(cfg, w16, w32), cfg


TEXT_SHAPES = {"vit-l-14": dict(vocab=49408, max_pos=77, hidden=768, layers=12, heads=12, ffn=3072, proj_dim=768),
               "vit-b-16": dict(vocab=49408, max_pos=77, hidden=512, layers=12, heads=8, ffn=2048, proj_dim=512)}


def random_clip_text(shape, seed=1, vocab=None):
    """The TEXT tower of the same CLIP model (vf_clip_text_*): ViT-L/14's is 12 layers x 768 wide, 77 positions, 768-d projection."""
    from veritasfi_amd import _ffi
    from veritasfi_amd.vision import HipClipTextEncoder
    import ctypes
    cfg = dict(TEXT_SHAPES[shape], act=1, eos_token_id=2, normalize=1, ln_eps=1e-5)
    if vocab:
        cfg["vocab"] = vocab
    c = _ffi.ClipTextConfig(**cfg)
    n16, n32 = _ffi.c_i64(0), _ffi.c_i64(0)
    _ffi.check(_ffi.lib().vf_clip_text_weight_sizes(ctypes.byref(c), ctypes.byref(n16), ctypes.byref(n32)))
    rng = np.random.default_rng(seed)
    w16 = (rng.standard_normal(n16.value, dtype=np.float32) * 0.03).astype(np.float16)
    w32 = np.zeros(n32.value, np.float32)
    H, F, L = cfg["hidden"], cfg["ffn"], cfg["layers"]
    per = 2 * H + 3 * H + H + 2 * H + F + H
    for l in range(L):
        o = l * per
        w32[o:o + H] = 1.0
        w32[o + 6 * H:o + 7 * H] = 1.0
    w32[L * per:L * per + H] = 1.0
    return HipClipTextEncoder(cfg, w16, w32), cfg


class ClipHashTokenizer:
    """CLIPTokenizer's call shape (padding="max_length", 77 positions, bos / eot, the eot id the LARGEST of the vocabulary so
    that the legacy argmax pooling finds it) over a whitespace hash -- no vocabulary files offline."""
    def __init__(self, vocab=49408):
        self.vocab, self.bos, self.eot = vocab, vocab - 2, vocab - 1

    def __call__(self, texts, padding="max_length", truncation=True, max_length=77, return_tensors="np", **_):
        texts = [texts] if isinstance(texts, str) else list(texts)
        ids = np.zeros((len(texts), max_length), np.int64)
        mask = np.zeros_like(ids)
        for i, tx in enumerate(texts):
            w = []
            for x in tx.split()[: max_length - 2]:
                h = 2166136261
                for ch in x.encode():
                    h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
                w.append(1 + h % (self.vocab - 3))
            row = [self.bos, *w, self.eot]
            ids[i, :len(row)], mask[i, :len(row)] = row, 1
        return {"input_ids": ids, "attention_mask": mask}


def flops_per_image(cfg):
    T = (cfg["image"] // cfg["patch"]) ** 2 + 1
    H, F, L = cfg["hidden"], cfg["ffn"], cfg["layers"]
    return L * (2.0 * T * (4 * H * H + 2 * H * F) + 4.0 * T * T * H) + 2.0 * (T - 1) * H * cfg["channels"] * cfg["patch"] ** 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="vit-l-14", choices=sorted(SHAPES))
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--u8", action="store_true", help="raw uint8 pixels, normalised on the device")
    a = ap.parse_args()
    enc, cfg = random_vit(a.shape)
    if a.u8:
        px = np.random.default_rng(1).integers(0, 256, (a.batch, 3, cfg["image"], cfg["image"]), dtype=np.uint8)
        fwd = enc.forward_u8
    else:
        px = np.random.default_rng(1).standard_normal((a.batch, 3, cfg["image"], cfg["image"]), dtype=np.float32)
        fwd = enc.forward
    fwd(px)
    ts = []
    for _ in range(a.iters):
        t0 = time.perf_counter()
        out = fwd(px)
        ts.append(time.perf_counter() - t0)
    assert np.isfinite(out).all()
    p50 = float(np.median(ts))
    print(json.dumps({"shape": a.shape, "batch": a.batch, "p50_ms": round(p50 * 1e3, 3), "images_per_s": round(a.batch / p50, 1),
                      "tflops_at_p50": round(flops_per_image(cfg) * a.batch / p50 / 1e12, 1),
                      "frac_of_2.5PF": round(flops_per_image(cfg) * a.batch / p50 / 2.5e15, 4), "pcie_inclusive": True, "pixels": "uint8" if a.u8 else "fp32"}))
    enc.close()


if __name__ == "__main__":
    main()
