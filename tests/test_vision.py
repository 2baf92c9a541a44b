"""The vision tower (CLIP-style ViT, BASELINE configs[3]'s "figure encoder") against transformers' CLIPVisionModelWithProjection
in fp32 -- the third-party model the config names; the reference itself holds no image model.  Weights are seeded random
initialisations (no checkpoints offline), inputs seeded normal "pixels"."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def vf():
    import veritasfi_amd as m
    from veritasfi_amd import _ffi
    _ffi.lib()
    return m


def _clip(hidden, layers, heads, ffn, image, patch, proj, act, seed=0):
    import torch
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    torch.manual_seed(seed)
    cfg = CLIPVisionConfig(hidden_size=hidden, intermediate_size=ffn, num_hidden_layers=layers, num_attention_heads=heads,
                           image_size=image, patch_size=patch, projection_dim=proj, hidden_act=act)
    m = CLIPVisionModelWithProjection(cfg).eval()
    with torch.no_grad():   # the default initialisation is tiny (std 0.02 * factor): give LayerNorms and biases something to do
        for n, p_ in m.named_parameters():
            if n.endswith("bias"):
                p_.normal_(0.0, 0.05)
            elif "layer_norm" in n or "layrnorm" in n or "layernorm" in n:
                p_.normal_(1.0, 0.1)
            elif p_.dim() >= 2:
                p_.normal_(0.0, 0.06)
    return m


def test_pack_layout_matches_the_header_sizes():
    """CPU: the packed blobs have exactly the sizes the header's layout prescribes (Kp = 3 * 8 * 8 = 192 here)."""
    from veritasfi_amd.vision import pack_hf_clip_vision
    m = _clip(128, 2, 2, 256, 32, 8, 64, "quick_gelu")
    cfg, w16, w32 = pack_hf_clip_vision(m)
    H, F, L, P, Kp, D = 128, 256, 2, 16, 192, 64
    assert cfg["act"] == 1 and cfg["proj_dim"] == D and cfg["image"] == 32 and cfg["patch"] == 8
    assert w16.size == H * Kp + H + (P + 1) * H + L * (3 * H * H + H * H + F * H + H * F) + D * H
    assert w32.size == 2 * H + L * (2 * H + 3 * H + H + 2 * H + F + H) + 2 * H
    assert w16.dtype == np.float16 and w32.dtype == np.float32


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [
    dict(hidden=128, layers=2, heads=2, ffn=256, image=32, patch=8, proj=64, act="quick_gelu", batch=5),      # 17 tokens -> 32
    dict(hidden=256, layers=3, heads=4, ffn=512, image=56, patch=14, proj=128, act="gelu", batch=9),          # patch length 588 -> 640, 17 tokens
    dict(hidden=768, layers=4, heads=12, ffn=3072, image=224, patch=16, proj=512, act="quick_gelu", batch=12),  # ViT-B/16 geometry: 197 tokens -> 224
    dict(hidden=1024, layers=2, heads=16, ffn=4096, image=224, patch=14, proj=768, act="quick_gelu", batch=6),  # ViT-L/14 geometry: 257 tokens -> 288, 768-d output
])
def test_vision_tower_matches_clip_fp32(vf, shape):
    import torch
    from veritasfi_amd.vision import HipVisionEncoder
    s = dict(shape)
    batch = s.pop("batch")
    m = _clip(**s)
    g = torch.Generator().manual_seed(11)
    px = torch.randn(batch, 3, s["image"], s["image"], generator=g)
    with torch.no_grad():
        want = m(pixel_values=px).image_embeds.numpy()
    enc = HipVisionEncoder.from_hf(m)
    try:
        got = enc.forward(px.numpy())
        again = enc.forward(px.numpy()[: max(1, batch // 2)])     # a smaller batch on the same handle: same rows
    finally:
        enc.close()
    assert got.shape == want.shape == (batch, s["proj"])
    scale = float(np.abs(want).max())
    err = float(np.abs(got - want).max()) / scale
    cos = float(np.min(np.sum(got * want, 1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))))
    print("vision tower", shape, "max err / max|embed|", err, "min cosine", cos)
    assert err < 6e-3 and cos > 0.9999          # fp16 operands and residual stream, fp32 accumulation: measured 1-2e-3
    assert np.array_equal(again, got[: again.shape[0]])


@pytest.mark.gpu
def test_image_embeddings_surface_and_errors(vf):
    import torch
    from veritasfi_amd.vision import HipImageEmbeddings, HipVisionEncoder
    m = _clip(128, 1, 2, 256, 32, 8, 64, "quick_gelu")
    enc = HipVisionEncoder.from_hf(m, normalize=True)
    emb = HipImageEmbeddings(enc, batch_size=4)
    px = torch.randn(10, 3, 32, 32, generator=torch.Generator().manual_seed(3)).numpy()
    out = emb.embed_images(px)
    assert len(out) == 10 and len(out[0]) == 64 and isinstance(out[0][0], float)
    assert np.allclose(np.linalg.norm(np.asarray(out), axis=1), 1.0, atol=1e-4)
    assert np.allclose(emb.embed_image(px[3]), out[3], atol=1e-6)
    assert emb.embed_images(px[:0]) == []
    # raw bytes, normalised on the device = the processor's rescale + normalize on the host, then the float entry
    from veritasfi_amd.vision import CLIP_MEAN, CLIP_STD
    u8 = np.random.default_rng(5).integers(0, 256, (6, 3, 32, 32), dtype=np.uint8)
    host = (u8.astype(np.float32) / 255.0 - np.asarray(CLIP_MEAN, np.float32)[None, :, None, None]) / np.asarray(CLIP_STD, np.float32)[None, :, None, None]
    a, b = enc.forward_u8(u8), enc.forward(host)
    assert np.abs(a - b).max() < 2e-3 and np.allclose(np.asarray(emb.embed_images(u8)), a, atol=1e-6)
    with pytest.raises(ValueError):
        enc.forward(px[:, :, :16])                 # wrong image size
    enc.close()
    with pytest.raises(RuntimeError):
        enc.forward(px)
