"""One-line construction from the reference's own configuration keys.

The reference builds its two model handles from names in ``config/example.yaml:1-15``::

    self.embeddings = HuggingFaceEmbeddings(model_name=self.embeddings_model_name)          # src/utils/ragManager.py:50
    self.reranker = FlagLLMReranker(config.get('rerank_model'), devices='cuda', use_fp16=True)   # src/utils/vllmChatService.py:90

Both third-party constructors take a hub name or a local directory and work the rest out from the files in it -- sentence-transformers
from ``modules.json`` / ``1_Pooling/config.json`` (CLS vs mean vs last token, a ``Normalize`` module, ``max_seq_length``), FlagEmbedding
from the model's architecture and its tokenizer's id of ``"Yes"``.  The functions here do the same and return the HIP-backed objects:

    emb = veritasfi_amd.HuggingFaceEmbeddings(model_name=cfg["embeddings_model_name"])     # -> HipEmbeddings / HipDecoderEmbeddings
    rr = veritasfi_amd.FlagLLMReranker(cfg["rerank_model"], devices="cuda", use_fp16=True)  # -> HipLLMReranker / HipReranker
    parts = veritasfi_amd.from_config("config/production.yaml")                             # both + the retriever class, from the YAML

``device_ids`` / ``corpus_dtype`` are the two optional keys SURVEY.md section 5 adds (existing YAMLs keep working without them):
several devices give one model replica per device (the reference's worker-per-GPU data parallelism, ``step3_mul.py:405-452``) and a
corpus sharded over them (``FaissRetriever(..., device_ids=[...])``); ``corpus_dtype`` in ``f32 | f16 | fp8`` is how the index holds
the rows in HBM.  Weights are read with ``transformers`` (fp32 on the host), rounded to fp16 once and handed to the library; nothing
here runs a forward on the CPU.
"""
from __future__ import annotations

import functools
import json
import os
import threading
import types

import numpy as np

from . import encoder as _enc

ENCODER_TYPES = ("bert", "roberta", "xlm-roberta")
DECODER_TYPES = ("qwen3", "gemma", "llama", "mistral", "qwen2")


def resolve_model_dir(path_or_name: str) -> str:
    """A local directory as it is; a hub name through ``huggingface_hub.snapshot_download`` (cache first; the download needs a
    network, exactly as the reference's constructors do)."""
    if os.path.isdir(path_or_name):
        return path_or_name
    try:
        from huggingface_hub import snapshot_download
        try:
            return snapshot_download(repo_id=path_or_name, local_files_only=True)
        except Exception:  # noqa: BLE001 -- not in the cache: try the hub
            return snapshot_download(repo_id=path_or_name)
    except Exception as e:  # noqa: BLE001
        raise FileNotFoundError(f"{path_or_name!r} is neither a local model directory nor a hub repository that can be fetched here "
                                f"({type(e).__name__}: {e})") from e


def read_sentence_transformers_layout(model_dir: str) -> dict:
    """What ``SentenceTransformer(model_dir)`` would assemble: {transformer_dir, pooling: 'cls' | 'mean' | 'lasttoken', normalize,
    max_seq_length | None, layout: 'modules.json' | 'plain'}.  A directory without ``modules.json`` is a plain transformers
    checkpoint: sentence-transformers then adds MEAN pooling over the unmasked tokens and no Normalize module."""
    mj = os.path.join(model_dir, "modules.json")
    if not os.path.exists(mj):
        return dict(transformer_dir=model_dir, pooling="mean", normalize=False, max_seq_length=None, layout="plain")
    modules = sorted(json.load(open(mj)), key=lambda m: m.get("idx", 0))
    out = dict(transformer_dir=model_dir, pooling=None, normalize=False, max_seq_length=None, layout="modules.json")
    for m in modules:
        kind, sub = m.get("type", ""), os.path.join(model_dir, m.get("path", ""))
        if kind.endswith(".Transformer"):
            out["transformer_dir"] = sub
            for name in ("sentence_bert_config.json", "sentence_roberta_config.json", "sentence_xlm-roberta_config.json"):
                f = os.path.join(sub, name)
                if os.path.exists(f):
                    out["max_seq_length"] = json.load(open(f)).get("max_seq_length")
                    break
        elif kind.endswith(".Pooling"):
            pc = json.load(open(os.path.join(sub, "config.json")))
            picked = [k for k in ("pooling_mode_cls_token", "pooling_mode_mean_tokens", "pooling_mode_lasttoken", "pooling_mode_max_tokens",
                                  "pooling_mode_mean_sqrt_len_tokens", "pooling_mode_weightedmean_tokens") if pc.get(k)]
            if len(picked) != 1 or picked[0] not in ("pooling_mode_cls_token", "pooling_mode_mean_tokens", "pooling_mode_lasttoken"):
                raise ValueError(f"{model_dir}: pooling {picked or pc} is not one of cls / mean / lasttoken")
            out["pooling"] = {"pooling_mode_cls_token": "cls", "pooling_mode_mean_tokens": "mean", "pooling_mode_lasttoken": "lasttoken"}[picked[0]]
        elif kind.endswith(".Normalize"):
            out["normalize"] = True
        else:
            raise ValueError(f"{model_dir}: sentence-transformers module {kind!r} has no HIP counterpart (Transformer, Pooling, Normalize do)")
    if out["pooling"] is None:
        raise ValueError(f"{model_dir}: modules.json names no Pooling module")
    return out


def _devices(device_id, device_ids):
    devs = [int(d) for d in device_ids] if device_ids else [int(device_id)]
    if not devs:
        raise ValueError("device_ids must name at least one device")
    return devs


class ReplicaSet:
    """One model replica per device behind a single object: batches (embedder) or pair blocks (re-ranker) go to the replicas in
    parallel, one host thread each, results come back in input order -- the reference's worker-per-GPU data parallelism
    (``experiments/retriever/step3_mul.py:38-58,405-452``) inside one process.  (``ShardedScorer`` is the one-process-per-GPU form.)"""

    def __init__(self, replicas):
        self.replicas = list(replicas)
        self._lock = threading.Lock()

    def _fan_out(self, n_items, work):
        n = len(self.replicas)
        bounds = [(n_items * r // n, n_items * (r + 1) // n) for r in range(n)]
        out, errs = [None] * n, []

        def run(r):
            lo, hi = bounds[r]
            try:
                out[r] = work(self.replicas[r], lo, hi) if hi > lo else []
            except BaseException as e:  # noqa: BLE001
                errs.append(e)
        with self._lock:
            ths = [threading.Thread(target=run, args=(r,)) for r in range(n)]
            [t.start() for t in ths]
            [t.join() for t in ths]
        if errs:
            raise errs[0]
        return [x for part in out for x in part]

    # embedder surface
    def embed_documents(self, texts):
        texts = list(texts)
        return self._fan_out(len(texts), lambda rep, lo, hi: rep.embed_documents(texts[lo:hi]))

    def embed_queries(self, texts):
        texts = list(texts)
        return self._fan_out(len(texts), lambda rep, lo, hi: rep.embed_queries(texts[lo:hi]))

    def embed_query(self, text):
        return self.replicas[0].embed_query(text)

    # re-ranker surface
    def compute_score(self, sentence_pairs, batch_size: int = 8, **kw):
        if len(sentence_pairs) and isinstance(sentence_pairs[0], str):
            sentence_pairs = [sentence_pairs]
        pairs = list(sentence_pairs)
        return self._fan_out(len(pairs), lambda rep, lo, hi: rep.compute_score(pairs[lo:hi], batch_size=batch_size, **kw))

    def close(self):
        for rep in self.replicas:
            for attr in ("encoder", "decoder"):
                h = getattr(rep, attr, None)
                if h is not None:
                    h.close()


def _max_length(tok, cfg, st_max, asked, position_room):
    lim = [int(x) for x in (asked, st_max, getattr(tok, "model_max_length", None), position_room) if x and int(x) < 10 ** 8]
    return min(lim) if lim else 512


def load_embeddings(model_name: str = None, device_id: int = 0, device_ids=None, batch_size: int = 32, max_length: int = None,
                    query_instruction: str = "", **_ignored):
    """``HuggingFaceEmbeddings(model_name=...)`` (ragManager.py:50): the embedder a sentence-transformers directory / hub name
    describes, on the HIP encoder.  BERT / RoBERTa / XLM-R encoders (bge, e5, MiniLM ...) give ``HipEmbeddings``; decoder embedders
    with last-token pooling (Qwen3-Embedding) give ``HipDecoderEmbeddings``.  Extra keyword arguments of the langchain class
    (``model_kwargs``, ``encode_kwargs``, ``cache_folder`` ...) are accepted and ignored."""
    if not model_name:
        raise ValueError("model_name is required")
    from transformers import AutoConfig, AutoModel, AutoTokenizer
    root = resolve_model_dir(model_name)
    lay = read_sentence_transformers_layout(root)
    tdir = lay["transformer_dir"]
    cfg = AutoConfig.from_pretrained(tdir)
    tok = AutoTokenizer.from_pretrained(tdir)
    devs = _devices(device_id, device_ids)
    if cfg.model_type in ENCODER_TYPES:
        if lay["pooling"] == "lasttoken":
            pooling = _enc.POOL_LAST_TOKEN
        else:
            pooling = _enc.POOL_CLS if lay["pooling"] == "cls" else _enc.POOL_MEAN_MASKED
        model = AutoModel.from_pretrained(tdir).eval()
        packed = _enc.pack_hf_weights(model, pooling, lay["normalize"])
        room = cfg.max_position_embeddings - ((cfg.pad_token_id + 1) if cfg.model_type in ("roberta", "xlm-roberta") else 0)
        ml = _max_length(tok, cfg, lay["max_seq_length"], max_length, room)
        reps = [_enc.HipEmbeddings(tok, _enc.HipEncoder(*packed, device_id=d), max_length=ml, batch_size=batch_size) for d in devs]
    elif cfg.model_type in DECODER_TYPES:
        if lay["pooling"] != "lasttoken":
            raise ValueError(f"{model_name}: a decoder embedder needs last-token pooling (its Pooling module selects {lay['pooling']})")
        model = AutoModel.from_pretrained(tdir).eval()
        packed = _enc.pack_hf_decoder_weights(model, _enc.POOL_LAST_TOKEN, lay["normalize"])
        ml = _max_length(tok, cfg, lay["max_seq_length"], max_length, _enc.DECODER_MAX_TOKENS)
        reps = [_enc.HipDecoderEmbeddings(tok, _enc.HipDecoder(*packed, device_id=d), max_length=ml, batch_size=min(batch_size, 16),
                                          query_instruction=query_instruction) for d in devs]
    else:
        raise ValueError(f"{model_name}: model type {cfg.model_type!r} has no HIP forward (supported: {ENCODER_TYPES + DECODER_TYPES})")
    emb = reps[0] if len(reps) == 1 else ReplicaSet(reps)
    emb.layout = lay
    return emb


def load_reranker(model_name_or_path: str, use_fp16: bool = True, devices=None, device_id: int = 0, device_ids=None, max_length: int = None,
                  prompt: str = None, **_ignored):
    """``FlagLLMReranker(config['rerank_model'], devices='cuda', use_fp16=True)`` (vllmChatService.py:90) and, for encoder
    cross-encoders, ``FlagReranker(name)``: ``compute_score(pairs, batch_size=8)`` on the HIP forward.  A decoder checkpoint
    (bge-reranker-v2-gemma: the configured one) scores the logit of its tokenizer's ``"Yes"`` at the last position
    (``HipLLMReranker``); an ``...ForSequenceClassification`` encoder (bge-reranker-base / -large) its single logit (``HipReranker``).
    ``use_fp16`` is what the library computes in anyway; ``devices`` ('cuda', 'cuda:1', [0, 1] ...) picks the device(s)."""
    from transformers import AutoConfig, AutoModelForCausalLM, AutoModelForSequenceClassification, AutoTokenizer
    if devices is not None and device_ids is None:
        dl = devices if isinstance(devices, (list, tuple)) else [devices]
        parsed = []
        for d in dl:
            if isinstance(d, int):
                parsed.append(d)
            elif isinstance(d, str) and d.startswith("cuda"):
                parsed.append(int(d.split(":")[1]) if ":" in d else 0)
            else:
                raise ValueError(f"devices={devices!r}: this library runs on MI355X GPUs ('cuda', 'cuda:N' or device numbers)")
        device_ids = parsed
    root = resolve_model_dir(model_name_or_path)
    cfg = AutoConfig.from_pretrained(root)
    tok = AutoTokenizer.from_pretrained(root)
    devs = _devices(device_id, device_ids)
    if cfg.model_type in ENCODER_TYPES:
        model = AutoModelForSequenceClassification.from_pretrained(root).eval()
        packed = _enc.pack_hf_weights(model)
        room = cfg.max_position_embeddings - ((cfg.pad_token_id + 1) if cfg.model_type in ("roberta", "xlm-roberta") else 0)
        ml = _max_length(tok, cfg, None, max_length or 512, room)
        reps = [_enc.HipReranker(tok, _enc.HipEncoder(*packed, device_id=d), max_length=ml) for d in devs]
    elif cfg.model_type in DECODER_TYPES:
        model = AutoModelForCausalLM.from_pretrained(root).eval()
        yes = tok("Yes", add_special_tokens=False)["input_ids"][0]         # FlagLLMReranker's yes_loc
        packed = _enc.pack_hf_decoder_weights(model, score_token=int(yes))
        reps = [_enc.HipLLMReranker(tok, _enc.HipDecoder(*packed, device_id=d), max_length=max_length or 1024, prompt=prompt) for d in devs]
        for r in reps:
            r.yes_loc = int(yes)
    else:
        raise ValueError(f"{model_name_or_path}: model type {cfg.model_type!r} has no HIP forward")
    return reps[0] if len(reps) == 1 else ReplicaSet(reps)


def HuggingFaceEmbeddings(model_name: str = None, **kw):
    """Same call as ``langchain_huggingface.HuggingFaceEmbeddings(model_name=...)`` (ragManager.py:8,50)."""
    return load_embeddings(model_name=model_name, **kw)


def FlagLLMReranker(model_name_or_path: str, **kw):
    """Same call as ``FlagEmbedding.FlagLLMReranker(name, devices='cuda', use_fp16=True)`` (vllmChatService.py:15,90)."""
    return load_reranker(model_name_or_path, **kw)


FlagReranker = FlagLLMReranker      # FlagEmbedding's encoder cross-encoder class: the same loader picks the head from the checkpoint


def load_config(cfg):
    """A dict as it is; a path through ``yaml.safe_load`` (``load_config``, src/load_data.py:19-21)."""
    if isinstance(cfg, (str, os.PathLike)):
        import yaml
        with open(cfg) as f:
            cfg = yaml.safe_load(f)
    if not isinstance(cfg, dict):
        raise TypeError("from_config takes the configuration dict or the path of its YAML file")
    return cfg


def from_config(cfg, load_models: bool = True):
    """The hot path's objects from the reference's configuration (``config/example.yaml:1-15``): ``.embeddings`` from
    ``embeddings_model_name``, ``.reranker`` from ``rerank_model``, ``.retriever_cls`` = ``FaissRetriever`` bound to the optional
    ``device_ids`` (default [0]) and ``corpus_dtype`` (default "f32": rows held as the caller's fp32, as faiss does), ``.rerank_topk``.
    ``load_models=False`` resolves the keys only (no weights are read)."""
    from .faiss_retriever import FaissRetriever
    cfg = load_config(cfg)
    for key in ("embeddings_model_name", "rerank_model"):
        if key not in cfg:
            raise KeyError(f"configuration has no {key!r} (config/example.yaml:3,9)")
    device_ids = cfg.get("device_ids", None)
    if device_ids is not None:
        device_ids = [int(d) for d in (device_ids if isinstance(device_ids, (list, tuple)) else [device_ids])]
    corpus_dtype = str(cfg.get("corpus_dtype", "f32")).lower()
    if corpus_dtype not in ("f32", "f16", "fp8"):
        raise ValueError(f"corpus_dtype {corpus_dtype!r}: one of f32, f16, fp8")
    first = device_ids[0] if device_ids else 0
    retriever_cls = functools.partial(FaissRetriever, device_id=first, device_ids=device_ids if device_ids and len(device_ids) > 1 else None,
                                      corpus_dtype=corpus_dtype)
    out = types.SimpleNamespace(config=cfg, device_ids=device_ids or [0], corpus_dtype=corpus_dtype, retriever_cls=retriever_cls,
                                rerank_topk=cfg.get("rerank_topk"), embeddings=None, reranker=None)
    if load_models:
        out.embeddings = load_embeddings(cfg["embeddings_model_name"], device_id=first, device_ids=device_ids)
        out.reranker = load_reranker(cfg["rerank_model"], device_id=first, device_ids=device_ids)
    return out
