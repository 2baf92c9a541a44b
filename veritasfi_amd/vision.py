"""CLIP-style towers on the GPU: the vision tower ("figure encoder" of BASELINE.json configs[3]) and, since round 4, the TEXT
tower of the same model -- the only way a text query reaches figure rows, which live in CLIP's joint space and not in the text
embedder's (``HipClipTextEncoder`` / ``HipClipTextEmbeddings`` at the end of this file; transformers'
``CLIPTextModelWithProjection`` is the contract: ``text_embeds = text_projection(final_layer_norm(h)[eos])``).

The reference has no image model to mirror (``grep -ri "clip\\|vit" /root/reference`` finds nothing); the contract is the
third-party model the config names, transformers' ``CLIPVisionModelWithProjection``:
``image_embeds = visual_projection(post_layernorm(last_hidden_state[:, 0]))``.  ``HipImageEmbeddings`` gives it the shape of
the text embedder next to it -- ``embed_images(pixel_values) -> list[list[float]]`` -- so that figure chunks land in the same
768-wide index as text chunks (ViT-L/14 projects to 768).  Pixels arrive already resized and normalised by the caller's image
processor (third-party preprocessing, like the tokenizers).  All arithmetic is HIP (``csrc/vf_transformer.hip``, ``vf_vit_*``);
there is no torch or CPU fallback.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _ffi
from .encoder import _np16, _np32

ACT_GELU, ACT_QUICK_GELU = 0, 1
CLIP_MEAN, CLIP_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)   # CLIPImageProcessor's defaults


def pack_hf_clip_vision(model, normalize=False):
    """Flatten a HF ``CLIPVisionModelWithProjection`` into the two blobs ``vf_vit_create`` takes (layout:
    include/veritasfi_hip.h).  Returns (cfg, w16, w32)."""
    sd = model.state_dict()
    c = model.config
    H, F, L = c.hidden_size, c.intermediate_size, c.num_hidden_layers
    act = {"gelu": ACT_GELU, "quick_gelu": ACT_QUICK_GELU}.get(c.hidden_act)
    if act is None:
        raise ValueError(f"activation {c.hidden_act!r} is not implemented (gelu, quick_gelu)")
    K = c.num_channels * c.patch_size * c.patch_size
    Kp = (K + 63) // 64 * 64
    pw = sd["vision_model.embeddings.patch_embedding.weight"].detach().cpu().float().numpy().reshape(H, K)
    patch = np.zeros((H, Kp), np.float16)
    patch[:, :K] = pw.astype(np.float16)
    w16 = [patch.ravel(), _np16(sd["vision_model.embeddings.class_embedding"]),
           _np16(sd["vision_model.embeddings.position_embedding.weight"])]
    # (the published checkpoints spell it "pre_layrnorm")
    w32 = [_np32(sd["vision_model.pre_layrnorm.weight"]), _np32(sd["vision_model.pre_layrnorm.bias"])]
    for l in range(L):
        p = f"vision_model.encoder.layers.{l}."
        w16 += [_np16(sd[p + "self_attn.q_proj.weight"]), _np16(sd[p + "self_attn.k_proj.weight"]),
                _np16(sd[p + "self_attn.v_proj.weight"]), _np16(sd[p + "self_attn.out_proj.weight"]),
                _np16(sd[p + "mlp.fc1.weight"]), _np16(sd[p + "mlp.fc2.weight"])]
        w32 += [_np32(sd[p + "layer_norm1.weight"]), _np32(sd[p + "layer_norm1.bias"]),
                _np32(sd[p + "self_attn.q_proj.bias"]), _np32(sd[p + "self_attn.k_proj.bias"]),
                _np32(sd[p + "self_attn.v_proj.bias"]), _np32(sd[p + "self_attn.out_proj.bias"]),
                _np32(sd[p + "layer_norm2.weight"]), _np32(sd[p + "layer_norm2.bias"]),
                _np32(sd[p + "mlp.fc1.bias"]), _np32(sd[p + "mlp.fc2.bias"])]
    w16 += [_np16(sd["visual_projection.weight"])]
    w32 += [_np32(sd["vision_model.post_layernorm.weight"]), _np32(sd["vision_model.post_layernorm.bias"])]
    cfg = dict(image=c.image_size, patch=c.patch_size, channels=c.num_channels, hidden=H, layers=L,
               heads=c.num_attention_heads, ffn=F, proj_dim=c.projection_dim, act=act, normalize=int(bool(normalize)),
               ln_eps=float(c.layer_norm_eps))
    return cfg, np.ascontiguousarray(np.concatenate(w16)), np.ascontiguousarray(np.concatenate(w32))


class HipVisionEncoder:
    """Handle over ``vf_vit_*``.  ``forward`` takes pixel_values [b, channels, image, image] fp32."""

    def __init__(self, cfg: dict, w16: np.ndarray, w32: np.ndarray, device_id: int = 0):
        L = _ffi.lib()
        self.cfg = dict(cfg)
        c = _ffi.VitConfig(**cfg)
        n16, n32 = _ffi.c_i64(0), _ffi.c_i64(0)
        _ffi.check(L.vf_vit_weight_sizes(ctypes.byref(c), ctypes.byref(n16), ctypes.byref(n32)), "vf_vit_weight_sizes")
        w16 = np.ascontiguousarray(w16, dtype=np.float16)
        w32 = np.ascontiguousarray(w32, dtype=np.float32)
        if w16.size != n16.value or w32.size != n32.value:
            raise ValueError(f"weight blobs have {w16.size}/{w32.size} elements, config needs {n16.value}/{n32.value}")
        self._h = _ffi.vp()
        _ffi.check(L.vf_vit_create(ctypes.byref(self._h), ctypes.byref(c), w16.ctypes.data, w16.size, w32.ctypes.data, w32.size,
                                   int(device_id)), "vf_vit_create")
        self.out_dim = int(cfg["proj_dim"])

    @classmethod
    def from_hf(cls, model, normalize=False, device_id: int = 0):
        return cls(*pack_hf_clip_vision(model, normalize), device_id=device_id)

    def forward(self, pixel_values) -> np.ndarray:
        if self._h is None:
            raise RuntimeError("HipVisionEncoder is closed")
        if hasattr(pixel_values, "detach"):
            pixel_values = pixel_values.detach().cpu().numpy()
        px = np.ascontiguousarray(pixel_values, dtype=np.float32)
        c = self.cfg
        if px.ndim != 4 or px.shape[1:] != (c["channels"], c["image"], c["image"]):
            raise ValueError(f"pixel_values must be [b, {c['channels']}, {c['image']}, {c['image']}], got {px.shape}")
        out = np.empty((px.shape[0], self.out_dim), np.float32)
        _ffi.check(_ffi.lib().vf_vit_forward(self._h, px.ctypes.data, px.shape[0], out.ctypes.data), "vf_vit_forward")
        return out

    def forward_u8(self, pixels_u8, mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
        """Raw bytes [b, 3, image, image] (already resized / cropped): the image processor's rescale (1 / 255) and normalize
        (mean, std) run on the device -- a quarter of the bytes over PCIe."""
        if self._h is None:
            raise RuntimeError("HipVisionEncoder is closed")
        px = np.ascontiguousarray(pixels_u8, dtype=np.uint8)
        c = self.cfg
        if px.ndim != 4 or px.shape[1:] != (3, c["image"], c["image"]) or c["channels"] != 3:
            raise ValueError(f"pixels_u8 must be [b, 3, {c['image']}, {c['image']}], got {px.shape}")
        m, s_ = np.ascontiguousarray(mean, np.float32), np.ascontiguousarray(std, np.float32)
        if m.shape != (3,) or s_.shape != (3,):
            raise ValueError("mean / std must have three entries")
        out = np.empty((px.shape[0], self.out_dim), np.float32)
        _ffi.check(_ffi.lib().vf_vit_forward_u8(self._h, px.ctypes.data, m.ctypes.data, s_.ctypes.data, px.shape[0], out.ctypes.data),
                   "vf_vit_forward_u8")
        return out

    def close(self):
        if getattr(self, "_h", None) is not None:
            _ffi.lib().vf_vit_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipImageEmbeddings:
    """``embed_images(pixel_values)`` / ``embed_image(one)`` with the text embedder's return types (lists of floats), in
    batches of ``batch_size`` images; ``image_processor`` (optional, HF-style callable returning ``pixel_values``) is applied
    to raw images first."""

    def __init__(self, encoder: HipVisionEncoder, image_processor=None, batch_size: int = 64, mean=CLIP_MEAN, std=CLIP_STD):
        self.encoder, self.image_processor, self.batch_size = encoder, image_processor, int(batch_size)
        self.mean, self.std = tuple(mean), tuple(std)

    def _pixels(self, images):
        if self.image_processor is not None:
            images = self.image_processor(images=images, return_tensors="np")["pixel_values"]
        a = np.asarray(images)
        return a if a.dtype == np.uint8 else a.astype(np.float32, copy=False)

    def embed_images(self, images):
        """float pixel_values (processor output) or, without a processor, raw uint8 [b, 3, image, image] (normalised on the
        device with ``mean`` / ``std``)."""
        px = self._pixels(images)
        fwd = (lambda x: self.encoder.forward_u8(x, self.mean, self.std)) if px.dtype == np.uint8 else self.encoder.forward
        out = [fwd(px[i:i + self.batch_size]) for i in range(0, len(px), self.batch_size)]
        return np.concatenate(out).tolist() if out else []

    def embed_image(self, image):
        return self.embed_images([image] if self.image_processor is not None else np.asarray(image)[None])[0]


# ---- text tower ---------------------------------------------------------------------------------------------------------
def pack_hf_clip_text(model, normalize=False):
    """Flatten a HF ``CLIPTextModelWithProjection`` into the two blobs ``vf_clip_text_create`` takes (layout:
    include/veritasfi_hip.h).  Returns (cfg, w16, w32)."""
    sd = model.state_dict()
    c = model.config
    H, F, L = c.hidden_size, c.intermediate_size, c.num_hidden_layers
    act = {"gelu": ACT_GELU, "quick_gelu": ACT_QUICK_GELU}.get(c.hidden_act)
    if act is None:
        raise ValueError(f"activation {c.hidden_act!r} is not implemented (gelu, quick_gelu)")
    w16 = [_np16(sd["text_model.embeddings.token_embedding.weight"]), _np16(sd["text_model.embeddings.position_embedding.weight"])]
    w32 = []
    for l in range(L):
        p = f"text_model.encoder.layers.{l}."
        w16 += [_np16(sd[p + "self_attn.q_proj.weight"]), _np16(sd[p + "self_attn.k_proj.weight"]),
                _np16(sd[p + "self_attn.v_proj.weight"]), _np16(sd[p + "self_attn.out_proj.weight"]),
                _np16(sd[p + "mlp.fc1.weight"]), _np16(sd[p + "mlp.fc2.weight"])]
        w32 += [_np32(sd[p + "layer_norm1.weight"]), _np32(sd[p + "layer_norm1.bias"]),
                _np32(sd[p + "self_attn.q_proj.bias"]), _np32(sd[p + "self_attn.k_proj.bias"]),
                _np32(sd[p + "self_attn.v_proj.bias"]), _np32(sd[p + "self_attn.out_proj.bias"]),
                _np32(sd[p + "layer_norm2.weight"]), _np32(sd[p + "layer_norm2.bias"]),
                _np32(sd[p + "mlp.fc1.bias"]), _np32(sd[p + "mlp.fc2.bias"])]
    w16 += [_np16(sd["text_projection.weight"])]
    w32 += [_np32(sd["text_model.final_layer_norm.weight"]), _np32(sd["text_model.final_layer_norm.bias"])]
    cfg = dict(vocab=c.vocab_size, max_pos=c.max_position_embeddings, hidden=H, layers=L, heads=c.num_attention_heads, ffn=F,
               proj_dim=c.projection_dim, act=act, eos_token_id=int(c.eos_token_id), normalize=int(bool(normalize)),
               ln_eps=float(c.layer_norm_eps))
    return cfg, np.ascontiguousarray(np.concatenate(w16)), np.ascontiguousarray(np.concatenate(w32))


class HipClipTextEncoder:
    """Handle over ``vf_clip_text_*``.  ``forward(input_ids, attention_mask=None)`` -> text_embeds [b, proj_dim] fp32."""

    def __init__(self, cfg: dict, w16: np.ndarray, w32: np.ndarray, device_id: int = 0):
        L = _ffi.lib()
        self.cfg = dict(cfg)
        c = _ffi.ClipTextConfig(**cfg)
        n16, n32 = _ffi.c_i64(0), _ffi.c_i64(0)
        _ffi.check(L.vf_clip_text_weight_sizes(ctypes.byref(c), ctypes.byref(n16), ctypes.byref(n32)), "vf_clip_text_weight_sizes")
        w16 = np.ascontiguousarray(w16, dtype=np.float16)
        w32 = np.ascontiguousarray(w32, dtype=np.float32)
        if w16.size != n16.value or w32.size != n32.value:
            raise ValueError(f"weight blobs have {w16.size}/{w32.size} elements, config needs {n16.value}/{n32.value}")
        self._h = _ffi.vp()
        _ffi.check(L.vf_clip_text_create(ctypes.byref(self._h), ctypes.byref(c), w16.ctypes.data, w16.size, w32.ctypes.data,
                                         w32.size, int(device_id)), "vf_clip_text_create")
        self.out_dim = int(cfg["proj_dim"])

    @classmethod
    def from_hf(cls, model, normalize=False, device_id: int = 0):
        return cls(*pack_hf_clip_text(model, normalize), device_id=device_id)

    def forward(self, input_ids, attention_mask=None) -> np.ndarray:
        if self._h is None:
            raise RuntimeError("HipClipTextEncoder is closed")
        if hasattr(input_ids, "detach"):
            input_ids = input_ids.detach().cpu().numpy()
        if hasattr(attention_mask, "detach"):
            attention_mask = attention_mask.detach().cpu().numpy()
        ids = np.ascontiguousarray(input_ids, dtype=np.int32)
        if ids.ndim != 2 or ids.shape[1] < 1 or ids.shape[1] > self.cfg["max_pos"]:
            raise ValueError(f"input_ids must be [b, 1..{self.cfg['max_pos']}], got {ids.shape}")
        mp = None
        if attention_mask is not None:
            m = np.ascontiguousarray(attention_mask, dtype=np.int32)
            if m.shape != ids.shape:
                raise ValueError("attention_mask must have the shape of input_ids")
            mp = m.ctypes.data
        out = np.empty((ids.shape[0], self.out_dim), np.float32)
        _ffi.check(_ffi.lib().vf_clip_text_forward(self._h, ids.ctypes.data, mp, ids.shape[0], ids.shape[1], out.ctypes.data),
                   "vf_clip_text_forward")
        return out

    def close(self):
        if getattr(self, "_h", None) is not None:
            _ffi.lib().vf_clip_text_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipClipTextEmbeddings:
    """The embedder surface of src/utils/ragManager.py:50 (``embed_query(str) -> list[float]``, ``embed_documents(list[str]) ->
    list[list[float]]``) over the CLIP text tower: what a figure-leg ``FaissRetriever(figure_rows, HipClipTextEmbeddings(...))``
    takes as its ``embedding_fn``.  ``tokenizer``: HF-style callable (``padding="max_length"``, ``truncation=True``,
    ``max_length``) returning ``input_ids`` and ``attention_mask``; tokenisation stays third-party, as for the text embedder."""

    def __init__(self, tokenizer, encoder: HipClipTextEncoder, batch_size: int = 256, pass_mask: bool = False):
        self.tokenizer, self.encoder, self.batch_size, self.pass_mask = tokenizer, encoder, int(batch_size), bool(pass_mask)
        self.max_length = int(encoder.cfg["max_pos"])

    def _embed(self, texts):
        out = []
        for i in range(0, len(texts), self.batch_size):
            enc = self.tokenizer(list(texts[i:i + self.batch_size]), padding="max_length", truncation=True, max_length=self.max_length,
                                 return_tensors="np")
            out.append(self.encoder.forward(enc["input_ids"], enc["attention_mask"] if self.pass_mask else None))
        return np.concatenate(out) if out else np.zeros((0, self.encoder.out_dim), np.float32)

    def embed_documents(self, texts):
        return self._embed(list(texts)).tolist()

    def embed_query(self, text):
        return self._embed([text])[0].tolist()
