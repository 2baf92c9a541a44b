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
    # a smaller batch may take other product kernels (the dispatch follows the tile count: round 4 moved mid-size products to the
    # 256 x 256 persistent kernel), i.e. another summation order: same rows to fp16 rounding, not to the bit
    assert float(np.abs(again - got[: again.shape[0]]).max()) / scale < 3e-3


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


# ---- CLIP TEXT tower (round 4): the query side of the figure leg ------------------------------------------------------------
def _clip_text(hidden, layers, heads, ffn, proj, act, vocab=1000, max_pos=77, eos=2, seed=1):
    import torch
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    torch.manual_seed(seed)
    cfg = CLIPTextConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=ffn, num_hidden_layers=layers,
                         num_attention_heads=heads, max_position_embeddings=max_pos, projection_dim=proj, hidden_act=act,
                         eos_token_id=eos, bos_token_id=1, pad_token_id=0)
    m = CLIPTextModelWithProjection(cfg).eval()
    with torch.no_grad():
        for n, p_ in m.named_parameters():
            if n.endswith("bias"):
                p_.normal_(0.0, 0.05)
            elif "layer_norm" in n:
                p_.normal_(1.0, 0.1)
            elif "embedding" in n:
                p_.normal_(0.0, 0.3)
            elif p_.dim() >= 2:
                p_.normal_(0.0, 0.06)
            p_.copy_(p_.half().float())
    return m


def _text_batch(rng, b, t, vocab, eos):
    """CLIP tokenizer shaped rows: bos, words, eos, then padding (the published tokenizer pads with the EOS id; with the legacy
    eos_token_id == 2 the pooled position is argmax(ids), so the real end-of-text id is the largest of the vocabulary)."""
    ids = np.zeros((b, t), np.int64)
    mask = np.zeros((b, t), np.int64)
    eot = vocab - 1 if eos == 2 else eos
    for i in range(b):
        n = t - 2 if i == 0 else int(rng.integers(1, t - 2))          # row 0 fills the window
        words = rng.integers(3, vocab - 1, n)
        if eos != 2:
            words[words == eos] = 3
        row = [1, *words.tolist(), eot]
        ids[i, :len(row)] = row
        ids[i, len(row):] = eot if eos != 2 else 0                     # pad: EOS id (new configs) or 0 (below every real id)
        mask[i, :len(row)] = 1
    return ids, mask


def test_text_pack_layout_matches_the_header_sizes():
    from veritasfi_amd.vision import pack_hf_clip_text
    m = _clip_text(128, 2, 2, 256, 64, "quick_gelu", vocab=300, max_pos=40)
    cfg, w16, w32 = pack_hf_clip_text(m)
    H, F, L, V, P, D = 128, 256, 2, 300, 40, 64
    assert cfg["act"] == 1 and cfg["proj_dim"] == D and cfg["eos_token_id"] == 2 and cfg["max_pos"] == P
    assert w16.size == V * H + P * H + L * (3 * H * H + H * H + F * H + H * F) + D * H
    assert w32.size == L * (2 * H + 3 * H + H + 2 * H + F + H) + 2 * H


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [
    dict(hidden=128, layers=2, heads=2, ffn=256, proj=64, act="quick_gelu", eos=2, batch=5, t=20, max_pos=40),
    dict(hidden=256, layers=3, heads=4, ffn=512, proj=128, act="gelu", eos=999, batch=9, t=77, max_pos=77),       # new-style eos id: first occurrence
    dict(hidden=512, layers=4, heads=8, ffn=2048, proj=512, act="quick_gelu", eos=2, batch=16, t=77, max_pos=77),  # ViT-B/32's text tower, 4 of 12 layers
    dict(hidden=768, layers=12, heads=12, ffn=3072, proj=768, act="quick_gelu", eos=2, batch=12, t=77, max_pos=77),  # ViT-L/14's text tower at full depth: 768-d joint space
])
def test_text_tower_matches_clip_fp32(vf, shape):
    """text_embeds of CLIPTextModelWithProjection (fp32, CPU, same fp16-rounded weights): causal attention, EOS pooling under both
    eos conventions, with and without the attention mask (the HF pipeline passes none), a smaller batch on the same handle."""
    import torch
    from veritasfi_amd.vision import HipClipTextEncoder
    s = dict(shape)
    batch, t, eos = s.pop("batch"), s.pop("t"), s["eos"]
    m = _clip_text(**s)
    rng = np.random.default_rng(17)
    ids, mask = _text_batch(rng, batch, t, 1000, eos)
    with torch.no_grad():
        want = m(input_ids=torch.from_numpy(ids)).text_embeds.numpy()
        want_masked = m(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).text_embeds.numpy()
    enc = HipClipTextEncoder.from_hf(m)
    try:
        got = enc.forward(ids)
        got_masked = enc.forward(ids, mask)
        again = enc.forward(ids[: max(1, batch // 2)])
    finally:
        enc.close()
    assert got.shape == want.shape == (batch, s["proj"])
    scale = float(np.abs(want).max())
    err = float(np.abs(got - want).max()) / scale
    err_m = float(np.abs(got_masked - want_masked).max()) / scale
    cos = float(np.min(np.sum(got * want, 1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))))
    print("text tower", shape, "max err / max|embed|", err, "masked", err_m, "min cosine", cos)
    assert err < 6e-3 and err_m < 6e-3 and cos > 0.9999      # the vision tower's tolerances (fp16 operands and residual stream)
    # a smaller batch may take other product kernels (the dispatch follows the tile count: round 4 moved mid-size products to the
    # 256 x 256 persistent kernel), i.e. another summation order: same rows to fp16 rounding, not to the bit
    assert float(np.abs(again - got[: again.shape[0]]).max()) / scale < 3e-3


class _WordTokenizer:
    """CLIPTokenizer-shaped stand-in (no vocabulary files offline): bos, one id per word, eot = the largest id, zero padding."""
    def __init__(self, vocab=1000):
        self.vocab = vocab

    def __call__(self, texts, padding="max_length", truncation=True, max_length=77, return_tensors="np"):
        ids = np.zeros((len(texts), max_length), np.int64)
        mask = np.zeros_like(ids)
        for i, tx in enumerate(texts):
            w = [3 + (sum(map(ord, x)) * 31 + len(x)) % (self.vocab - 5) for x in tx.split()][: max_length - 2]
            row = [1, *w, self.vocab - 1]
            ids[i, :len(row)], mask[i, :len(row)] = row, 1
        return {"input_ids": ids, "attention_mask": mask}


@pytest.mark.gpu
def test_clip_text_embeddings_surface_and_errors(vf):
    """embed_query / embed_documents (the ragManager.py:50 embedder surface) over the text tower, and a figure leg served by it:
    FaissRetriever(figure_rows, HipClipTextEmbeddings).invoke([text]) searches CLIP space with a CLIP query vector."""
    from veritasfi_amd.vision import HipClipTextEmbeddings, HipClipTextEncoder
    m = _clip_text(128, 2, 2, 256, 64, "quick_gelu")
    enc = HipClipTextEncoder.from_hf(m, normalize=True)
    emb = HipClipTextEmbeddings(_WordTokenizer(), enc, batch_size=3)
    docs = [f"figure {i} quarterly deliveries by region chart {i % 4}" for i in range(8)]
    vecs = np.asarray(emb.embed_documents(docs), np.float32)
    assert vecs.shape == (8, 64) and np.allclose(np.linalg.norm(vecs, axis=1), 1.0, atol=1e-4)
    q = emb.embed_query(docs[5])
    assert isinstance(q, list) and isinstance(q[0], float) and np.allclose(q, vecs[5], atol=1e-6)
    assert emb.embed_documents([]) == []
    fr = vf.FaissRetriever(vecs.tolist(), emb)                      # rows in CLIP space, queries through the CLIP text tower
    I, D = fr.invoke([docs[2], docs[7]], 3)
    assert I[0, 0] == 2 and I[1, 0] == 7 and np.all(D[:, 0] > 0.999)
    with pytest.raises(ValueError):
        enc.forward(np.zeros((2, 78), np.int64))                    # beyond max_pos
    for bad in (-1, 1000, 1 << 20):                                 # an id outside the token table is refused by name, not clamped
        ids = np.ones((2, 8), np.int64)
        ids[1, 3] = bad
        with pytest.raises(RuntimeError, match="outside the vocabulary"):
            enc.forward(ids)
    assert np.isfinite(enc.forward(np.ones((2, 8), np.int64))).all()   # the handle survives the refusals
    enc.close()
    with pytest.raises(RuntimeError):
        enc.forward(np.zeros((1, 8), np.int64))
