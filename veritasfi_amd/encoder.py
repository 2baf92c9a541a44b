"""Encoder forward on the GPU: the embedding model and the cross-encoder re-ranker.

Drop-in objects for the reference's third-party model handles:

* ``HipEmbeddings`` -- ``.embed_query(str)`` / ``.embed_documents(list[str])``, the surface of
  ``HuggingFaceEmbeddings`` constructed at ``src/utils/ragManager.py:50`` and used at
  ``src/utils/faissRetriever.py:33``, ``src/utils/ensembleRetriever.py:248,275`` and (through
  ``Chroma.add_texts``) by the embed loop ``src/load_data.py:98-99,120-128``.
* ``HipReranker`` -- ``.compute_score(pairs, batch_size=8) -> list[float]`` (raw logits), the call at
  ``src/utils/vllmManager.py:450-452``; encoder-style cross-encoder (XLM-R + classification head), the
  model family BASELINE.json configs[3]/[4] name.
* ``HipModel`` -- callable with the HF signature returning ``.last_hidden_state`` so that
  ``get_embeddings`` (``experiments/retriever/step3_mul.py:191-209``) runs its own pooling on it.

Weights come from any HF BERT / RoBERTa / XLM-R module (``pack_hf_weights``); tokenisation stays in Python
(any HF-style tokenizer callable).  All arithmetic is HIP (``csrc/vf_transformer.hip``) behind
``vf_encoder_*`` / ``vf_reranker_*``; there is no torch or CPU fallback.
"""
from __future__ import annotations

import ctypes
import types

import numpy as np

from . import _ffi
from .host_tokenize import BatchTokenizer, pipelined, split_for_overlap

POOL_CLS, POOL_MEAN_UNMASKED, POOL_LAST_TOKEN, POOL_MEAN_MASKED = 0, 1, 2, 3   # 3: sentence-transformers' mean over the unmasked tokens


def _np16(t):
    return t.detach().cpu().float().numpy().astype(np.float16).ravel()


def _np32(t):
    return t.detach().cpu().float().numpy().astype(np.float32).ravel()


def pack_hf_weights(model, pooling=POOL_CLS, normalize=True):
    """Flatten a HF ``BertModel`` / ``RobertaModel`` / ``XLMRobertaModel`` or ``...ForSequenceClassification``
    into the two blobs ``vf_encoder_create`` takes (layout: include/veritasfi_hip.h).  Returns (cfg, w16, w32)."""
    sd = model.state_dict()
    cfgm = model.config
    prefix = ""
    for cand in ("roberta.", "bert.", ""):
        if f"{cand}embeddings.word_embeddings.weight" in sd:
            prefix = cand
            break
    has_head = "classifier.out_proj.weight" in sd
    is_roberta = cfgm.model_type in ("roberta", "xlm-roberta")
    H, F, L = cfgm.hidden_size, cfgm.intermediate_size, cfgm.num_hidden_layers
    if has_head and sd["classifier.out_proj.weight"].shape[0] != 1:
        raise ValueError("only single-logit classification heads are supported (num_labels=1)")
    if getattr(cfgm, "hidden_act", "gelu") != "gelu":
        raise ValueError("only exact GELU is implemented")
    g = lambda k: sd[prefix + k]
    w16 = [_np16(g("embeddings.word_embeddings.weight")), _np16(g("embeddings.position_embeddings.weight")),
           _np16(g("embeddings.token_type_embeddings.weight"))]
    w32 = [_np32(g("embeddings.LayerNorm.weight")), _np32(g("embeddings.LayerNorm.bias"))]
    for l in range(L):
        p = f"encoder.layer.{l}."
        w16 += [_np16(g(p + "attention.self.query.weight")), _np16(g(p + "attention.self.key.weight")),
                _np16(g(p + "attention.self.value.weight")), _np16(g(p + "attention.output.dense.weight")),
                _np16(g(p + "intermediate.dense.weight")), _np16(g(p + "output.dense.weight"))]
        w32 += [_np32(g(p + "attention.self.query.bias")), _np32(g(p + "attention.self.key.bias")),
                _np32(g(p + "attention.self.value.bias")), _np32(g(p + "attention.output.dense.bias")),
                _np32(g(p + "attention.output.LayerNorm.weight")), _np32(g(p + "attention.output.LayerNorm.bias")),
                _np32(g(p + "intermediate.dense.bias")), _np32(g(p + "output.dense.bias")),
                _np32(g(p + "output.LayerNorm.weight")), _np32(g(p + "output.LayerNorm.bias"))]
    if has_head:
        w16 += [_np16(sd["classifier.dense.weight"]), _np16(sd["classifier.out_proj.weight"])]
        w32 += [_np32(sd["classifier.dense.bias"]), _np32(sd["classifier.out_proj.bias"])]
    cfg = dict(vocab=cfgm.vocab_size, hidden=H, layers=L, heads=cfgm.num_attention_heads, ffn=F,
               max_pos=cfgm.max_position_embeddings, type_vocab=cfgm.type_vocab_size,
               roberta_pad_idx=(cfgm.pad_token_id if is_roberta else -1), pooling=pooling,
               normalize=int(bool(normalize)) if not has_head else 0, head=int(has_head),
               ln_eps=float(cfgm.layer_norm_eps))
    return cfg, np.ascontiguousarray(np.concatenate(w16)), np.ascontiguousarray(np.concatenate(w32))


class HipEncoder:
    """Handle over ``vf_encoder_*``.  ``forward`` takes int token ids / mask [b, t] (any t <= 8192 within the model's position table)."""

    def __init__(self, cfg: dict, w16: np.ndarray, w32: np.ndarray, device_id: int = 0):
        L = _ffi.lib()
        self.cfg = dict(cfg)
        c = _ffi.EncoderConfig(**cfg)
        n16, n32 = _ffi.c_i64(0), _ffi.c_i64(0)
        _ffi.check(L.vf_encoder_weight_sizes(ctypes.byref(c), ctypes.byref(n16), ctypes.byref(n32)), "vf_encoder_weight_sizes")
        w16 = np.ascontiguousarray(w16, dtype=np.float16)
        w32 = np.ascontiguousarray(w32, dtype=np.float32)
        if w16.size != n16.value or w32.size != n32.value:
            raise ValueError(f"weight blobs have {w16.size}/{w32.size} elements, config needs {n16.value}/{n32.value}")
        self._h = _ffi.vp()
        create = L.vf_reranker_create if cfg.get("head", 0) == 1 else L.vf_encoder_create
        _ffi.check(create(ctypes.byref(self._h), ctypes.byref(c), w16.ctypes.data, w16.size, w32.ctypes.data, w32.size,
                          int(device_id)), "vf_encoder_create")
        self.hidden = int(cfg["hidden"])
        self.out_dim = 1 if cfg.get("head", 0) == 1 else self.hidden

    @classmethod
    def from_hf(cls, model, pooling=POOL_CLS, normalize=True, device_id: int = 0):
        return cls(*pack_hf_weights(model, pooling, normalize), device_id=device_id)

    @staticmethod
    def _pad(ids, mask, type_ids):
        ids = np.asarray(ids, dtype=np.int32)
        mask = np.asarray(mask, dtype=np.int32)
        b, t = ids.shape
        tp = max(32, -(-t // 32) * 32)
        if tp > 8192:
            raise ValueError("sequences longer than 8192 tokens are not supported")
        def pad(a):
            if a is None:
                return None
            out = np.zeros((b, tp), dtype=np.int32)
            out[:, :t] = np.asarray(a, dtype=np.int32)
            return out
        return pad(ids), pad(mask), pad(type_ids), b, t, tp

    def forward(self, ids, mask, type_ids=None, pooling=None, normalize=None) -> np.ndarray:
        """Pooled output: [b, hidden] embeddings, or [b] logits for a re-ranker.  ``pooling`` / ``normalize``
        override the handle's setting for this call (embedding handles only)."""
        ids, mask, tt, b, t, tp = self._pad(ids, mask, type_ids)
        out = np.empty((b, self.out_dim), dtype=np.float32)
        ptt = tt.ctypes.data if tt is not None else None
        if pooling is not None or normalize is not None:
            rc = _ffi.lib().vf_encoder_forward_pooled(self._h, ids.ctypes.data, mask.ctypes.data, ptt, b, tp, t,
                                                      -1 if pooling is None else int(pooling),
                                                      -1 if normalize is None else int(bool(normalize)), out.ctypes.data)
        elif self.cfg.get("head", 0) == 1:
            rc = _ffi.lib().vf_reranker_score(self._h, ids.ctypes.data, mask.ctypes.data, ptt, b, tp, out.ctypes.data)
        else:
            rc = _ffi.lib().vf_encoder_forward(self._h, ids.ctypes.data, mask.ctypes.data, ptt, b, tp, t, out.ctypes.data)
        _ffi.check(rc, "vf_encoder_forward")
        return out[:, 0] if self.out_dim == 1 else out

    def hidden_states(self, ids, mask, type_ids=None) -> np.ndarray:
        """last_hidden_state [b, t, hidden] fp32 (padding columns beyond the given t removed)."""
        ids, mask, tt, b, t, tp = self._pad(ids, mask, type_ids)
        out = np.empty((b, tp, self.hidden), dtype=np.float32)
        _ffi.check(_ffi.lib().vf_encoder_forward_hidden(self._h, ids.ctypes.data, mask.ctypes.data,
                                                        tt.ctypes.data if tt is not None else None, b, tp,
                                                        out.ctypes.data), "vf_encoder_forward_hidden")
        return out[:, :t]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _ffi.lib().vf_encoder_destroy(self._h)
            self._h = _ffi.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _tok_arrays(enc):
    """ids / mask / token_type_ids as numpy from whatever an HF-style tokenizer returned."""
    def arr(x):
        if x is None:
            return None
        if hasattr(x, "detach"):
            x = x.detach().cpu().numpy()
        return np.asarray(x)
    return arr(enc["input_ids"]), arr(enc["attention_mask"]), arr(enc.get("token_type_ids") if hasattr(enc, "get") else None)


class HipModel:
    """HF-signature callable: ``model(**inputs).last_hidden_state`` (torch tensor), for get_embeddings."""

    def __init__(self, encoder: HipEncoder):
        self.encoder = encoder

    def __call__(self, input_ids=None, attention_mask=None, token_type_ids=None, **_):
        import torch
        ids, mask, tt = _tok_arrays({"input_ids": input_ids, "attention_mask": attention_mask,
                                     "token_type_ids": token_type_ids})
        return types.SimpleNamespace(last_hidden_state=torch.from_numpy(self.encoder.hidden_states(ids, mask, tt)))

    def pooled(self, pooling: str, input_ids=None, attention_mask=None, token_type_ids=None, **_):
        """get_embeddings fast path: forward + the caller's pooling ("last_token" | "mean") on the GPU, un-normalised,
        -> float32 [b, hidden] ndarray; skips the [b, t, hidden] copy-back and the CPU pooling."""
        ids, mask, tt = _tok_arrays({"input_ids": input_ids, "attention_mask": attention_mask,
                                     "token_type_ids": token_type_ids})
        return self.encoder.forward(ids, mask, tt, pooling={"last_token": POOL_LAST_TOKEN, "mean": POOL_MEAN_UNMASKED}[pooling],
                                    normalize=False)


class HipEmbeddings:
    """``HuggingFaceEmbeddings``-shaped embedder.  ``tokenizer(texts, padding=True, truncation=True,
    max_length=..., return_tensors="np")`` must return input_ids / attention_mask."""

    def __init__(self, tokenizer, encoder: HipEncoder, max_length: int = 512, batch_size: int = 32, overlap_tokenize: bool = True):
        self.tokenizer, self.encoder = tokenizer, encoder
        self.max_length, self.batch_size = max_length, batch_size
        # batch i + 1 is tokenised (Rust backend of a fast tokenizer, host_tokenize.py) while batch i is on the device and batch i - 1's
        # result is converted to lists.  The partition does not depend on the flag: same bits either way.
        self.overlap_tokenize = overlap_tokenize
        self._tok = BatchTokenizer(tokenizer, max_length)

    @classmethod
    def from_pretrained(cls, model_name: str, **kw):
        """``HuggingFaceEmbeddings(model_name=...)`` (src/utils/ragManager.py:50) in one line: a sentence-transformers directory or hub
        name -> tokenizer + weights + pooling / Normalize read from ``modules.json`` / ``1_Pooling/config.json`` (pretrained.py).
        A decoder embedder (last-token pooling) comes back as ``HipDecoderEmbeddings``; ``device_ids=[...]`` gives a replica per GPU."""
        from .pretrained import load_embeddings
        return load_embeddings(model_name, **kw)

    def _embed(self, texts, as_lists: bool = False):
        if not texts:
            return [] if as_lists else np.zeros((0, self.encoder.hidden), np.float32)
        if self._tok.max_length != self.max_length or self._tok.tokenizer is not self.tokenizer:
            self._tok = BatchTokenizer(self.tokenizer, self.max_length)
        # device batches of exactly batch_size texts, as before round 6: an embedding's last bits follow the batch it was computed in (the
        # product kernels' tile count), and a text must embed to the same bits whether it arrives in a call of 100 or of 1000 -- so a
        # single-batch call is NOT cut in halves here (the re-ranker's scores carry no such identity: HipReranker does cut)
        step = max(1, int(self.batch_size))
        pieces = [(lo, min(lo + step, len(texts))) for lo in range(0, len(texts), step)]
        out = pipelined(pieces, lambda p: self._tok.encode(texts[p[0]:p[1]]), lambda a: self.encoder.forward(*a),
                        self.overlap_tokenize and self._tok.direct,     # (a pure-Python tokenizer holds the interpreter lock: serial loop)
                        finish=(lambda a: [row.tolist() for row in a]) if as_lists else None)   # (row by row: switch points for the driving thread)
        if as_lists:     # list[list[float]], what the langchain interface returns; each piece was converted beside the next piece's forward
            return [row for piece in out for row in piece]
        return np.vstack(out)

    def embed_documents(self, texts):
        return self._embed(list(texts), as_lists=True)

    def embed_query(self, text):
        return self._embed([text])[0].tolist()

    def embed_queries(self, texts):
        """Batched queries (one forward) -- used by FaissRetriever.invoke instead of len(texts) forwards."""
        return self._embed(list(texts), as_lists=True)


class HipReranker:
    """``reranker.compute_score(pairs, batch_size=8)`` (vllmManager.py:451): raw logits, one per pair.

    ``batch_size`` is the caller's memory knob upstream (8 pairs per forward on a 24 GB card); a pair's score does
    not depend on what it is batched with, so by default consecutive micro-batches are fused into forwards of up to
    ``max_batch_tokens`` padded tokens (one 100 x 512 forward takes half the time of thirteen 8 x 512 ones).  Pass
    ``fuse_batches=False`` to run exactly ``batch_size`` pairs per forward."""

    def __init__(self, tokenizer, encoder: HipEncoder, max_length: int = 512, fuse_batches: bool = True,
                 max_batch_tokens: int = 65536, overlap_tokenize: bool = True):
        if encoder.cfg.get("head", 0) != 1:
            raise ValueError("HipReranker needs an encoder built from a sequence-classification model")
        self.tokenizer, self.encoder, self.max_length = tokenizer, encoder, max_length
        self.fuse_batches, self.max_batch_tokens = fuse_batches, max_batch_tokens
        self.overlap_tokenize = overlap_tokenize       # tokenise piece i + 1 under piece i's forward (host_tokenize.py)
        self._toks = {}

    @classmethod
    def from_pretrained(cls, model_name_or_path: str, **kw):
        """``FlagReranker(name)`` / ``FlagLLMReranker(name, devices='cuda', use_fp16=True)`` (src/utils/vllmChatService.py:90) in one
        line (pretrained.py): an encoder cross-encoder checkpoint gives a ``HipReranker``, a decoder one a ``HipLLMReranker``."""
        from .pretrained import load_reranker
        return load_reranker(model_name_or_path, **kw)

    def _tok_for(self, max_length: int) -> BatchTokenizer:
        tk = self._toks.get(max_length)
        if tk is None or tk.tokenizer is not self.tokenizer:
            tk = self._toks[max_length] = BatchTokenizer(self.tokenizer, max_length)
        return tk

    def compute_score(self, sentence_pairs, batch_size: int = 8, max_length: int = None, normalize: bool = False):
        if len(sentence_pairs) and isinstance(sentence_pairs[0], str):
            sentence_pairs = [sentence_pairs]
        max_length = max_length or self.max_length
        step = int(batch_size)
        if self.fuse_batches:
            step = max(step, (self.max_batch_tokens // max_length) // step * step)
        tk = self._tok_for(max_length)
        # device batches of `step` pairs; with fused batches a call that fits ONE of them (the reference's 100 pairs) goes as two
        # halves so that the second is tokenised while the first runs.  fuse_batches=False keeps exactly batch_size pairs per forward.
        # A tokenizer that runs in Python (no Rust backend) holds the interpreter lock while it works: a prefetch thread would make the
        # thread that drives the device wait for it (up to the 5-ms switch interval) instead of hiding it -- such a tokenizer keeps whole
        # batches and the serial loop.  The partition depends on the tokenizer's kind, never on overlap_tokenize.
        halves = self.fuse_batches and tk.direct
        pieces = split_for_overlap(len(sentence_pairs), step) if halves else \
            [(lo, min(lo + step, len(sentence_pairs))) for lo in range(0, len(sentence_pairs), step)]
        outs = pipelined(pieces, lambda p: tk.encode([q for q, _ in sentence_pairs[p[0]:p[1]]], [d for _, d in sentence_pairs[p[0]:p[1]]]),
                         lambda a: self.encoder.forward(*a), self.overlap_tokenize and tk.direct)
        scores = [float(v) for s_ in outs for v in np.atleast_1d(s_)]
        if normalize:
            scores = [1.0 / (1.0 + np.exp(-v)) for v in scores]
        return scores


# ---- decoder-only models (Qwen3-Embedding, "Yes"-logit LLM re-rankers) -----------------------------------------------
def pack_hf_decoder_weights(model, pooling=POOL_LAST_TOKEN, normalize=False, score_token=None, lm_head=None):
    """HF ``Qwen3Model`` / ``GemmaModel`` (or the ``...ForCausalLM`` with ``score_token``: head 2, the logit of that
    vocabulary token at the last position -- stress_test.py:197,212-225; ``bge-reranker-v2-gemma`` of config/example.yaml:9
    is a gemma) -> (cfg dict, fp16 blob, fp32 blob) in include/veritasfi_hip.h order."""
    core = getattr(model, "model", model)
    c = core.config
    head_dim = getattr(c, "head_dim", None) or c.hidden_size // c.num_attention_heads
    rope_theta = getattr(c, "rope_theta", None)
    if rope_theta is None:
        rope_theta = (getattr(c, "rope_parameters", None) or getattr(c, "rope_scaling", None) or {}).get("rope_theta", 10000.0)
    act_name = getattr(c, "hidden_activation", None) or getattr(c, "hidden_act", "silu")
    if act_name not in ("silu", "gelu_pytorch_tanh"):
        raise ValueError(f"unsupported gated-MLP activation {act_name}")
    gemma = getattr(c, "model_type", "") == "gemma"     # zero-centred norm gains, embeddings times sqrt(hidden)
    cfg = dict(vocab=c.vocab_size, hidden=c.hidden_size, layers=c.num_hidden_layers, heads=c.num_attention_heads,
               kv_heads=c.num_key_value_heads, head_dim=head_dim, ffn=c.intermediate_size, rope_theta=float(rope_theta),
               rms_eps=float(c.rms_norm_eps), qk_norm=int(hasattr(core.layers[0].self_attn, "q_norm")), pooling=int(pooling),
               normalize=int(bool(normalize)), head=2 if score_token is not None else 0,
               act=int(act_name == "gelu_pytorch_tanh"), norm_plus_one=int(gemma),
               embed_scale=float(c.hidden_size ** 0.5) if gemma else 1.0)
    p16, p32 = [_np16(core.embed_tokens.weight)], []
    for lyr in core.layers:
        a, m = lyr.self_attn, lyr.mlp
        for lin in (a.q_proj, a.k_proj, a.v_proj, a.o_proj):
            if getattr(lin, "bias", None) is not None:
                raise ValueError("attention projections with bias are not supported")
        p16 += [_np16(a.q_proj.weight), _np16(a.k_proj.weight), _np16(a.v_proj.weight), _np16(a.o_proj.weight),
                _np16(m.gate_proj.weight), _np16(m.up_proj.weight), _np16(m.down_proj.weight)]
        ones = np.ones(head_dim, np.float32)
        p32 += [_np32(lyr.input_layernorm.weight), _np32(lyr.post_attention_layernorm.weight),
                _np32(a.q_norm.weight) if cfg["qk_norm"] else ones, _np32(a.k_norm.weight) if cfg["qk_norm"] else ones]
    p32.append(_np32(core.norm.weight))
    if score_token is not None:
        head = lm_head if lm_head is not None else getattr(model, "lm_head", None)
        w = core.embed_tokens.weight if head is None else head.weight     # tied embeddings when there is no lm_head
        p16.append(_np16(w[int(score_token)]))
    return cfg, np.concatenate([a.ravel() for a in p16]), np.concatenate([a.ravel() for a in p32])


DECODER_MAX_TOKENS = 4096   # the reference's truncation length (experiments/retriever/step3_mul.py:200), kDecMaxT in csrc


class HipDecoder:
    """Decoder-only encoder / scorer on the GPU (``vf_decoder_*``).  ``forward(ids, mask)`` -> [b, hidden] pooled
    embeddings (head 0) or [b] logits of the scored token (head 2)."""

    def __init__(self, cfg: dict, w16: np.ndarray, w32: np.ndarray, device_id: int = 0):
        self.cfg = dict(cfg)
        self._cfg = _ffi.DecoderConfig(**cfg)
        self._h = _ffi.vp()
        w16 = np.ascontiguousarray(w16, dtype=np.float16)
        w32 = np.ascontiguousarray(w32, dtype=np.float32)
        _ffi.check(_ffi.lib().vf_decoder_create(ctypes.byref(self._h), ctypes.byref(self._cfg), w16.ctypes.data, w16.size,
                                                w32.ctypes.data, w32.size, int(device_id)), "vf_decoder_create")
        self.hidden = int(cfg["hidden"])
        self.out_dim = 1 if cfg.get("head", 0) == 2 else self.hidden

    @classmethod
    def from_hf(cls, model, pooling=POOL_LAST_TOKEN, normalize=False, score_token=None, device_id: int = 0):
        return cls(*pack_hf_decoder_weights(model, pooling, normalize, score_token), device_id=device_id)

    def forward(self, ids, mask, normalize=None) -> np.ndarray:
        """normalize: None = as the handle was created; True / False = L2-normalise the pooled rows (or not) for this call, in the
        pooling kernel (``vf_decoder_forward_pooled``)."""
        ids = np.asarray(ids, dtype=np.int32)
        mask = np.asarray(mask, dtype=np.int32)
        b, t = ids.shape
        tp = max(32, -(-t // 32) * 32)
        if tp > DECODER_MAX_TOKENS:
            raise ValueError(f"sequences longer than {DECODER_MAX_TOKENS} tokens are not supported")
        if tp != t:     # alignment columns on the right with mask 0 (t_valid tells the pooling where the tokenizer stopped)
            pi, pm = np.zeros((b, tp), np.int32), np.zeros((b, tp), np.int32)
            pi[:, :t], pm[:, :t] = ids, mask
            ids, mask = pi, pm
        ids, mask = np.ascontiguousarray(ids), np.ascontiguousarray(mask)
        out = np.empty((b, self.out_dim), dtype=np.float32)
        if normalize is None:
            _ffi.check(_ffi.lib().vf_decoder_forward(self._h, ids.ctypes.data, mask.ctypes.data, b, tp, t, out.ctypes.data),
                       "vf_decoder_forward")
        else:
            _ffi.check(_ffi.lib().vf_decoder_forward_pooled(self._h, ids.ctypes.data, mask.ctypes.data, b, tp, t, int(bool(normalize)),
                                                            out.ctypes.data), "vf_decoder_forward_pooled")
        return out[:, 0] if self.out_dim == 1 else out

    def hidden_states(self, ids, mask) -> np.ndarray:
        """last_hidden_state [b, t, hidden] float32 (after the final RMSNorm), as HF's ``model(**inputs)`` returns it."""
        ids = np.asarray(ids, dtype=np.int32)
        mask = np.asarray(mask, dtype=np.int32)
        b, t = ids.shape
        tp = max(32, -(-t // 32) * 32)
        if tp > DECODER_MAX_TOKENS:
            raise ValueError(f"sequences longer than {DECODER_MAX_TOKENS} tokens are not supported")
        if tp != t:
            pi, pm = np.zeros((b, tp), np.int32), np.zeros((b, tp), np.int32)
            pi[:, :t], pm[:, :t] = ids, mask
            ids, mask = pi, pm
        ids, mask = np.ascontiguousarray(ids), np.ascontiguousarray(mask)
        out = np.empty((b, tp, self.hidden), dtype=np.float32)
        _ffi.check(_ffi.lib().vf_decoder_forward_hidden(self._h, ids.ctypes.data, mask.ctypes.data, b, tp, out.ctypes.data),
                   "vf_decoder_forward_hidden")
        return out[:, :t]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _ffi.lib().vf_decoder_destroy(self._h)
            self._h = _ffi.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipDecoderModel:
    """HF-signature callable for get_embeddings: ``model(**inputs).last_hidden_state`` (the reference's generic route,
    step3_mul.py:203-207) and the ``pooled`` fast path (last_token_pool on the GPU, no [b, t, hidden] copy-back)."""

    def __init__(self, decoder: HipDecoder):
        self.decoder = decoder

    def pooled(self, pooling: str, input_ids=None, attention_mask=None, **_):
        if pooling != "last_token":
            raise ValueError("HipDecoderModel pools with last_token_pool (step3_mul.py:181-188)")
        ids, mask, _ = _tok_arrays({"input_ids": input_ids, "attention_mask": attention_mask, "token_type_ids": None})
        return self.decoder.forward(ids, mask)

    def __call__(self, input_ids=None, attention_mask=None, **_):
        import torch
        ids, mask, _tt = _tok_arrays({"input_ids": input_ids, "attention_mask": attention_mask, "token_type_ids": None})
        return types.SimpleNamespace(last_hidden_state=torch.from_numpy(self.decoder.hidden_states(ids, mask)))


class HipDecoderEmbeddings:
    """``HuggingFaceEmbeddings``-shaped embedder over a decoder-only model (``embed_query`` / ``embed_documents`` /
    ``embed_queries``): last_token_pool + L2 normalisation, left padding (the reference's tokenizer setting for its
    decoder embedder, continuous_retrieval.py:55-60).  ``query_instruction`` is prepended to queries only (instruction-tuned
    embedders such as Qwen3-Embedding expect one); documents are embedded as given."""

    def __init__(self, tokenizer, decoder: HipDecoder, max_length: int = 512, batch_size: int = 16, query_instruction: str = "",
                 overlap_tokenize: bool = True):
        if decoder.cfg.get("head", 0) != 0 or decoder.cfg.get("pooling", 2) != 2:
            raise ValueError("HipDecoderEmbeddings needs a decoder built with pooling=2 (last token) and no scoring head")
        self.tokenizer, self.decoder = tokenizer, decoder
        self.max_length, self.batch_size, self.query_instruction = max_length, batch_size, query_instruction
        self.pad_id = getattr(tokenizer, "pad_token_id", None) or 0
        self.overlap_tokenize = overlap_tokenize
        self._tok = BatchTokenizer(tokenizer, max_length)

    def _tokenize(self, texts):
        rows = self._tok.encode_plain(texts, self.max_length, add_special_tokens=True)
        width = max(len(r) for r in rows)
        ids = np.full((len(rows), width), self.pad_id, np.int32)
        mask = np.zeros((len(rows), width), np.int32)
        for j, r in enumerate(rows):                      # left padding: every row ends in a real token
            ids[j, width - len(r):] = r
            mask[j, width - len(r):] = 1
        return ids, mask

    def _embed(self, texts):
        if not texts:
            return np.zeros((0, self.decoder.hidden), np.float32)
        if self._tok.tokenizer is not self.tokenizer:
            self._tok = BatchTokenizer(self.tokenizer, self.max_length)
        step = max(1, int(self.batch_size))
        pieces = [(lo, min(lo + step, len(texts))) for lo in range(0, len(texts), step)]     # (whole batches: see HipEmbeddings._embed)
        # L2 normalisation in the pooling kernel (vf_decoder_forward_pooled), whatever the handle was created with
        out = pipelined(pieces, lambda p: self._tokenize(texts[p[0]:p[1]]), lambda a: self.decoder.forward(a[0], a[1], normalize=True),
                        self.overlap_tokenize and self._tok.rust_backed)
        return np.vstack(out)

    def embed_documents(self, texts):
        return self._embed(list(texts)).tolist()

    def embed_query(self, text):
        return self._embed([self.query_instruction + text])[0].tolist()

    def embed_queries(self, texts):
        return self._embed([self.query_instruction + t for t in texts]).tolist()


DEFAULT_RERANK_PROMPT = ("Given a query A and a passage B, determine whether the passage contains an answer to the query by "
                         "providing a prediction of either 'Yes' or 'No'.")


def build_llm_reranker_inputs(pairs, tokenizer, prompt=None, max_length=1024, batch_tokenizer=None):
    """Token ids of the LLM re-ranker's inputs, as ``get_inputs`` builds them (experiments/profile/stress_test.py:97-134;
    FlagLLMReranker): ``[bos] + tok("A: " + query)`` (query truncated to 3/4 of max_length) followed by
    ``tok("\n") + tok("B: " + passage)`` with ONLY THE SECOND part truncated so that the pair fits max_length, then
    ``tok("\n") + tok(prompt)``.  Returns a list of id lists (unpadded).  The queries and the passages of a call are each tokenised
    in ONE batch (host_tokenize.BatchTokenizer: the Rust backend of a fast tokenizer; per text otherwise)."""
    prompt = DEFAULT_RERANK_PROMPT if prompt is None else prompt
    bt = batch_tokenizer if batch_tokenizer is not None else BatchTokenizer(tokenizer, max_length)
    prompt_ids, sep_ids = bt.encode_plain([prompt, "\n"])
    pairs = list(pairs)
    q_all = bt.encode_plain([f"A: {query}" for query, _ in pairs], max_length * 3 // 4)
    p_all = bt.encode_plain([f"B: {passage}" for _, passage in pairs], max_length)
    out = []
    for q_ids, p_ids in zip(q_all, p_all):
        first = [tokenizer.bos_token_id] + q_ids
        second = sep_ids + p_ids
        room = max_length - len(first)
        if len(second) > room:                       # truncation='only_second'
            second = second[:max(room, 0)]
        out.append(first + second + sep_ids + prompt_ids)
    return out


class HipLLMReranker:
    """``compute_score(pairs, batch_size=8)`` of a decoder-only re-ranker (vllmManager.py:450-452 with the configured
    ``bge-reranker-v2-gemma``-style scorer): the raw logit of the "Yes" token at the last position of each prompt-wrapped
    pair.  ``decoder`` is a ``HipDecoder`` built with ``score_token=<id of "Yes">``."""

    def __init__(self, tokenizer, decoder: HipDecoder, max_length: int = 1024, prompt: str = None, fuse_batches: bool = True,
                 max_batch_tokens: int = 32768, overlap_tokenize: bool = True):
        if decoder.cfg.get("head", 0) != 2:
            raise ValueError("HipLLMReranker needs a decoder built with score_token=...")
        self.tokenizer, self.decoder, self.max_length, self.prompt = tokenizer, decoder, max_length, prompt
        self.fuse_batches, self.max_batch_tokens = fuse_batches, max_batch_tokens
        self.pad_id = getattr(tokenizer, "pad_token_id", None) or 0
        self.overlap_tokenize = overlap_tokenize
        self._tok = BatchTokenizer(tokenizer, max_length)

    from_pretrained = HipReranker.from_pretrained      # (the loader picks the class from the checkpoint)

    def _pad_left(self, batch):
        width = -(-max(len(r) for r in batch) // 8) * 8          # pad_to_multiple_of=8, LEFT padding: the score is read
        ids = np.full((len(batch), width), self.pad_id, np.int32)  # at the last column (logits[:, -1, yes_loc])
        mask = np.zeros((len(batch), width), np.int32)
        for j, r in enumerate(batch):
            ids[j, width - len(r):] = r
            mask[j, width - len(r):] = 1
        return ids, mask

    def compute_score(self, sentence_pairs, batch_size: int = 8, max_length: int = None, normalize: bool = False):
        if len(sentence_pairs) and isinstance(sentence_pairs[0], str):
            sentence_pairs = [sentence_pairs]
        max_length = max_length or self.max_length
        if self._tok.tokenizer is not self.tokenizer:
            self._tok = BatchTokenizer(self.tokenizer, self.max_length)
        n = len(sentence_pairs)
        step = int(batch_size)
        if self.fuse_batches and n:
            # device batches sized from the longest POSSIBLE row (max_length + prompt), so that the partition is known before
            # anything is tokenised and piece i + 1 can be tokenised under piece i's forward
            step = max(step, (self.max_batch_tokens // max(max_length, 1)) // step * step)
        pieces = split_for_overlap(n, step, min_piece=8) if (self.fuse_batches and self._tok.rust_backed) else \
            [(lo, min(lo + step, n)) for lo in range(0, n, step)]

        def prepare(p):
            return self._pad_left(build_llm_reranker_inputs(sentence_pairs[p[0]:p[1]], self.tokenizer, self.prompt, max_length, self._tok))
        outs = pipelined(pieces, prepare, lambda a: self.decoder.forward(a[0], a[1]), self.overlap_tokenize and self._tok.rust_backed)
        scores = [float(v) for s_ in outs for v in np.atleast_1d(s_)]
        if normalize:
            scores = [1.0 / (1.0 + np.exp(-v)) for v in scores]
        return scores
