"""GPU parity of the HIP encoder forward (embedding model + cross-encoder re-ranker) against the same
architecture in plain PyTorch fp32 on the CPU (HF BertModel / XLMRobertaForSequenceClassification with
seeded random weights: no checkpoints exist in this pipeline).  Floating-point path: fp16 weights and
activations with fp32 accumulation, so the bar is a tolerance, written in each test."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ROOT = ROOT
# Decoder-family tolerances (fp16 operands, fp32 residual stream, against HF fp32 on the CPU), from the values every
# decoder test measured on an MI355X (profiles/r03_decoder_errors.jsonl: largest 1 - cos 7.2e-7, largest relative error of a
# last-token embedding 1.35e-3, largest "Yes"-logit error 7.8e-4 on logits of magnitude 0.3 - 0.8): ~2-4x those.
DEC_COS_TOL, DEC_REL_TOL, DEC_LOGIT_TOL = 3e-6, 3e-3, 2.5e-3
# full-depth models (18 / 36 layers at the real widths): ~3x what the first GPU run measured (profiles/r04_decoder_errors.jsonl)
# measured: gemma-2b geometry 1 - cos 9.5e-7, relative error 1.5e-3, "Yes"-logit error 2.9e-3 on logits of magnitude 1.9 (1.5e-3
# relative); Qwen3-Embedding-4B geometry 1 - cos 4.5e-6, relative error 2.5e-3
FULL_GEMMA_COS_TOL, FULL_GEMMA_REL_TOL, FULL_GEMMA_LOGIT_TOL = 3e-6, 4.5e-3, 5e-3
FULL_QWEN_COS_TOL, FULL_QWEN_REL_TOL = 1.5e-5, 7.5e-3


@pytest.fixture(scope="module")
def vf():
    import veritasfi_amd as m
    from veritasfi_amd import _ffi
    _ffi.lib()
    return m


def _hf_bert(hidden, layers, heads, ffn, vocab=1000, seed=0):
    import torch
    from transformers import BertConfig, BertModel
    torch.manual_seed(seed)
    cfg = BertConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=ffn,
                     vocab_size=vocab, max_position_embeddings=512)
    m = BertModel(cfg, add_pooling_layer=False).eval()
    m = m.half().float()  # both sides see the same fp16-representable weights
    return m


def _hf_xlmr_cls(hidden, layers, heads, ffn, vocab=1200, seed=1):
    import torch
    from transformers import XLMRobertaConfig, XLMRobertaForSequenceClassification
    torch.manual_seed(seed)
    cfg = XLMRobertaConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                           intermediate_size=ffn, vocab_size=vocab, max_position_embeddings=514, type_vocab_size=1,
                           num_labels=1, pad_token_id=1)
    m = XLMRobertaForSequenceClassification(cfg).eval()
    for p in m.parameters():  # random-init heads are tiny; give the logits some spread
        if p.dim() == 1:
            p.data.add_(0.05 * torch.randn_like(p))
    return m.half().float()


def _batch(rng, b, t, vocab, pad_id=0, ragged=True):
    ids = rng.integers(5, vocab, size=(b, t)).astype(np.int64)
    mask = np.ones((b, t), dtype=np.int64)
    if ragged:
        for i in range(b):
            n = int(rng.integers(max(2, t // 4), t + 1)) if i else t
            mask[i, n:] = 0
            ids[i, n:] = pad_id
    return ids, mask


def _assert_rank_order(ref, got, top=None):
    """What a score error of e = max |ref - got| can and cannot do to a ranking.  e < half the smallest reference gap: the orders
    are identical, asserted outright.  Otherwise two items may trade places only if their reference scores are closer than 2 e --
    every other pair keeps its order, and (``top``) the top-``top`` SETS agree up to items within 2 e of the cut.  (The earlier form,
    `same order OR e < min gap`, let an error between gap / 2 and gap flip a pair and pass.)  Returns (e, min gap, discordant pairs)."""
    ref, got = np.asarray(ref, np.float64), np.asarray(got, np.float64)
    e = float(np.abs(ref - got).max())
    gap = float(np.min(np.diff(np.sort(ref)))) if ref.size > 1 else np.inf
    if e < 0.5 * gap:
        assert np.array_equal(np.argsort(-ref, kind="stable"), np.argsort(-got, kind="stable")), "rank order differs although the error is below half the smallest gap"
    dr, dg = ref[:, None] - ref[None, :], got[:, None] - got[None, :]
    disc = np.argwhere((dr > 0) & (dg < 0))
    for i, j in disc:
        assert dr[i, j] < 2 * e, f"items {i}, {j}: reference gap {dr[i, j]:.3e} > 2 x the largest score error {e:.3e}, yet their order flipped"
    if top is not None and top < ref.size:
        cut = np.sort(ref)[::-1][top - 1:top + 1].mean()
        sure_in, sure_out = ref > cut + 2 * e, ref < cut - 2 * e
        chosen = np.zeros(ref.size, bool)
        chosen[np.argsort(-got, kind="stable")[:top]] = True
        assert chosen[sure_in].all() and not chosen[sure_out].any(), "the top set differs outside the band the score error explains"
    return e, gap, len(disc)


@pytest.mark.parametrize("name,hidden,layers,heads,ffn,b,t", [
    ("tiny", 128, 2, 2, 512, 3, 48),
    ("small-odd-t", 256, 3, 4, 1024, 5, 100),
    ("bge-base-shape", 768, 12, 12, 3072, 4, 128),
    ("single-query", 768, 3, 12, 3072, 1, 40),     # <= 64 tokens with ffn >= 2048: the split-K FFN-down kernel
    ("large-batch", 768, 2, 12, 3072, 32, 512),    # 16384 rows: every GEMM takes the 128x256 DMA / 16x16x32 kernel
])
def test_embedding_encoder_matches_torch_fp32(vf, name, hidden, layers, heads, ffn, b, t):
    import torch
    model = _hf_bert(hidden, layers, heads, ffn)
    rng = np.random.default_rng(3)
    ids, mask = _batch(rng, b, t, 1000)
    with torch.no_grad():
        ref_h = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).last_hidden_state.numpy()
    ref = ref_h[:, 0]
    ref = ref / np.linalg.norm(ref, axis=1, keepdims=True)
    enc = vf.HipEncoder.from_hf(model, pooling=0, normalize=True)
    got = enc.forward(ids, mask)
    hs = enc.hidden_states(ids, mask)
    enc.close()
    assert got.shape == (b, hidden) and hs.shape == (b, t, hidden)
    cos = np.sum(got * ref, axis=1)
    err = np.abs(got - ref).max()
    herr = np.abs(hs - ref_h)[mask.astype(bool)]
    print(name, "cos", cos.min(), "max|d emb|", err, "hidden mean/max err", herr.mean(), herr.max())
    # tolerance: fp16 activations through `layers` post-LN blocks; LN outputs are O(1).  Measured on MI355X (round 1):
    # max |d emb| 2.3e-4, hidden mean 1.0e-3 / max 1.3e-2 at 12 layers -- asserted at ~3x that
    assert cos.min() > 0.99999 and err < 8e-4
    assert herr.mean() < 3e-3 and herr.max() < 4e-2


def test_pooling_variants(vf):
    import torch
    from veritasfi_amd.retrieval import get_embeddings
    model = _hf_bert(128, 2, 2, 512)
    rng = np.random.default_rng(4)
    ids, mask = _batch(rng, 4, 64, 1000)
    with torch.no_grad():
        h = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).last_hidden_state
    cfg, w16, w32 = vf.pack_hf_weights(model, pooling=1, normalize=False)   # unmasked mean (continuous_retrieval.py:148)
    e = vf.HipEncoder(cfg, w16, w32)
    # padded columns: the reference averages over whatever the tokenizer padded to; here T = 64 exactly
    assert np.abs(e.forward(ids, mask) - h.mean(dim=1).numpy()).max() < 2e-2
    e.close()
    cfg["pooling"] = 2                                                      # last_token_pool (step3_mul.py:181-188)
    e = vf.HipEncoder(cfg, w16, w32)
    lens = mask.sum(1) - 1
    want = h[torch.arange(4), torch.from_numpy(lens)].numpy()              # right padding branch
    assert np.abs(e.forward(ids, mask) - want).max() < 3e-2
    full = np.ones_like(mask)
    with torch.no_grad():
        h2 = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(full)).last_hidden_state
    assert np.abs(e.forward(ids, full) - h2[:, -1].numpy()).max() < 3e-2   # "left padding" branch: [:, -1]
    # the reference-shaped get_embeddings running its own pooling on HipModel's hidden states
    class Tok:
        def __call__(self, texts, padding=True, truncation=True, return_tensors="pt", max_length=None):
            rows = [ids[int(x)] for x in texts]
            m = [mask[int(x)] for x in texts]
            return {"input_ids": torch.from_numpy(np.stack(rows)), "attention_mask": torch.from_numpy(np.stack(m))}
    emb = get_embeddings(["0", "1", "2", "3"], vf.HipModel(e), Tok(), "cpu", batch_size=2, pooling="last_token")
    assert emb.shape == (4, 128) and np.abs(emb - want).max() < 3e-2
    # get_embeddings took HipModel.pooled (forward + pooling in one GPU call); the generic route -- an HF-style
    # callable whose hidden states the reference's own pooling code reduces on the host -- must agree with it,
    # for both poolings, whatever the handle was built with (per-call override, vf_encoder_forward_pooled)
    class HiddenOnly:
        def __init__(self, m): self.m = m
        def __call__(self, **kw): return self.m(**kw)
    for pooling, ref in (("last_token", want), ("mean", h.mean(dim=1).numpy())):
        fast = get_embeddings(["0", "1", "2", "3"], vf.HipModel(e), Tok(), "cpu", batch_size=4, pooling=pooling)
        slow = get_embeddings(["0", "1", "2", "3"], HiddenOnly(vf.HipModel(e)), Tok(), "cpu", batch_size=4, pooling=pooling)
        assert np.abs(fast - slow).max() < 2e-3 and np.abs(fast - ref).max() < 3e-2
    assert np.abs(e.forward(ids, mask) - want).max() < 3e-2                 # the handle's own setting is untouched
    with pytest.raises(RuntimeError):
        e.forward(ids, mask, pooling=5)
    e.close()


@pytest.mark.parametrize("hidden,layers,heads,ffn,b,t", [(128, 2, 2, 512, 6, 64), (768, 12, 12, 3072, 8, 256)])
def test_reranker_matches_torch_fp32(vf, hidden, layers, heads, ffn, b, t):
    import torch
    model = _hf_xlmr_cls(hidden, layers, heads, ffn)
    rng = np.random.default_rng(5)
    ids, mask = _batch(rng, b, t, 1200, pad_id=1)
    with torch.no_grad():
        ref = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).logits.view(-1).numpy()
    rr = vf.HipEncoder.from_hf(model)
    got = rr.forward(ids, mask)
    rr.close()
    print("reranker logits", ref[:4], got[:4], np.abs(ref - got).max())
    assert got.shape == (b,)
    assert np.abs(ref - got).max() < 1e-3 * max(1.0, np.abs(ref).max())   # measured 7e-4; the bar of DESIGN.md section 2 (logits: 1e-3 relative)
    _assert_rank_order(ref, got)


@pytest.mark.parametrize("kind,hidden,layers,heads,ffn,b,t,ragged", [
    ("embedder", 256, 4, 4, 1024, 16, 128, True),
    ("embedder-768", 768, 3, 12, 3072, 8, 256, True),       # three column tiles of row sums per row
    ("reranker", 256, 3, 4, 1024, 24, 96, True),            # 2304 rows: packed forward (ragged, CLS head) on the folded path
    ("reranker-full", 512, 2, 8, 2048, 8, 256, False),
])
def test_layernorm_folded_into_the_products(vf, kind, hidden, layers, heads, ffn, b, t, ragged):
    """LayerNorm without a launch (LnFold): with every product of a layer on the 8-phase kernel, the residual products leave
    raw sums + row sums, the next product reads them through gamma-folded weights and the next residual product normalises
    its residual element by element (an option: measured slower than the launches it removes).  Against HF
    fp32 at the tolerances of the unfolded path, and against the unfolded path itself; vf_debug_ln_fold_forwards proves the folded path ran (the tile threshold is lowered for the small model)."""
    import ctypes
    import torch
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    L.vf_debug_gemm_8p_min_wgs.restype = ctypes.c_longlong
    L.vf_debug_gemm_8p_min_wgs.argtypes = [ctypes.c_longlong]
    L.vf_debug_ln_fold_forwards.restype = ctypes.c_longlong
    was_on = L.vf_debug_ln_fold(1)                         # measured slower than the two launches it replaces: off by default
    rng = np.random.default_rng(21)
    head = kind.startswith("reranker")
    model = _hf_xlmr_cls(hidden, layers, heads, ffn) if head else _hf_bert(hidden, layers, heads, ffn)
    with torch.no_grad():                                  # LayerNorm gains / shifts away from (1, 0): the fold must carry them
        for name, p_ in model.named_parameters():
            if "LayerNorm" in name:
                p_.add_(0.2 * torch.randn_like(p_))
                p_.copy_(p_.half().float())
    ids, mask = _batch(rng, b, t, 1200 if head else 1000, pad_id=1 if head else 0, ragged=ragged)
    tid, tm = torch.from_numpy(ids), torch.from_numpy(mask)
    with torch.no_grad():
        if head:
            ref = model(input_ids=tid, attention_mask=tm).logits.view(-1).numpy()
        else:
            ref_h = model(input_ids=tid, attention_mask=tm).last_hidden_state.numpy()
            ref = ref_h[:, 0] / np.linalg.norm(ref_h[:, 0], axis=1, keepdims=True)
    enc = vf.HipEncoder.from_hf(model) if head else vf.HipEncoder.from_hf(model, pooling=0, normalize=True)
    prev = L.vf_debug_gemm_8p_min_wgs(1)
    try:
        n0 = L.vf_debug_ln_fold_forwards()
        got = enc.forward(ids, mask)
        hs = None if head else enc.hidden_states(ids, mask)
        assert L.vf_debug_ln_fold_forwards() >= n0 + 1, "the forward did not take the folded-LayerNorm path"
        L.vf_debug_gemm_8p_min_wgs(1 << 40)                # the same forward on the launch-per-LayerNorm path
        n1 = L.vf_debug_ln_fold_forwards()
        plain = enc.forward(ids, mask)
        assert L.vf_debug_ln_fold_forwards() == n1
    finally:
        L.vf_debug_gemm_8p_min_wgs(prev)
        L.vf_debug_ln_fold(was_on)
        enc.close()
    err, dplain = float(np.abs(got - ref).max()), float(np.abs(got - plain).max())
    print("ln-fold", kind, "max|d| vs HF fp32", err, "vs the unfolded path", dplain, "unfolded vs HF", float(np.abs(plain - ref).max()))
    if head:
        assert err < 2.5e-3 * max(1.0, np.abs(ref).max()) and dplain < 2.5e-3 * max(1.0, np.abs(ref).max())
    else:
        herr = np.abs(hs - ref_h)[mask.astype(bool)]
        print("   hidden mean / max err", herr.mean(), herr.max())
        assert err < 8e-4 and dplain < 8e-4 and herr.mean() < 3e-3 and herr.max() < 4e-2


def test_drop_in_embedder_and_reranker_objects(vf):
    """HipEmbeddings / HipReranker carry the reference's method names and feed FaissRetriever / rank fusion."""
    import torch
    rng = np.random.default_rng(6)

    class Tok:  # whitespace-hash tokenizer with the HF call signature (no tokenizer files offline)
        def __call__(self, a, b=None, padding=True, truncation=True, max_length=64, return_tensors="np"):
            a = [a] if isinstance(a, str) else list(a)
            b = [None] * len(a) if b is None else list(b)
            rows = []
            for x, y in zip(a, b):
                toks = [2] + [5 + (hash(w) % 900) for w in x.split()] + [3]
                if y is not None:
                    toks += [5 + (hash(w) % 900) for w in y.split()] + [3]
                rows.append(toks[:max_length])
            t = max(len(r) for r in rows)
            ids = np.ones((len(rows), t), np.int64)
            mask = np.zeros((len(rows), t), np.int64)
            for i, r in enumerate(rows):
                ids[i, :len(r)] = r
                mask[i, :len(r)] = 1
            return {"input_ids": ids, "attention_mask": mask}

    emb = vf.HipEmbeddings(Tok(), vf.HipEncoder.from_hf(_hf_bert(128, 2, 2, 512)), max_length=64)
    docs = [f"doc number {i} about topic {i % 7} and item {i * 3}" for i in range(300)]
    vecs = emb.embed_documents(docs)
    assert len(vecs) == 300 and len(vecs[0]) == 128 and abs(np.linalg.norm(vecs[5]) - 1.0) < 1e-3
    q = emb.embed_query(docs[17])
    assert np.abs(np.asarray(q) - np.asarray(vecs[17])).max() < 2e-3      # batch-size independence
    fr = vf.FaissRetriever(vecs, emb)                                      # the reference's constructor call shape
    I, D = fr.invoke([docs[17], docs[250]], 10)
    assert I[0, 0] == 17 and I[1, 0] == 250 and D[0, 0] > 0.999
    mtx = vf.compute_similarity_mtx(emb, docs[:12])
    assert tuple(mtx.shape) == (12, 12) and float(mtx[3, 3]) > 0.999
    rr = vf.HipReranker(Tok(), vf.HipEncoder.from_hf(_hf_xlmr_cls(128, 2, 2, 512)), max_length=64)
    pairs = [["what is topic 3", d] for d in docs[:20]]
    s8 = rr.compute_score(pairs, batch_size=8)
    s20 = rr.compute_score(pairs, batch_size=20)
    assert len(s8) == 20 and all(isinstance(v, float) for v in s8)
    assert np.abs(np.asarray(s8) - np.asarray(s20)).max() < 5e-3           # micro-batching does not change scores
    scores, order = vf.fuse_and_rank(s8, [0.1] * 20)
    assert sorted(order) == list(range(20)) and scores[order[0]] == max(scores)


def test_encoder_errors(vf):
    model = _hf_bert(128, 1, 2, 512)
    cfg, w16, w32 = vf.pack_hf_weights(model)
    with pytest.raises(ValueError):
        vf.HipEncoder(cfg, w16[:-1], w32)
    bad = dict(cfg, hidden=96)
    with pytest.raises((RuntimeError, ValueError)):
        vf.HipEncoder(bad, w16, w32)
    e = vf.HipEncoder(cfg, w16, w32)
    with pytest.raises(RuntimeError, match="position table"):       # 600 tokens against a 512-entry position table
        e.forward(np.zeros((1, 600), np.int64), np.ones((1, 600), np.int64))
    with pytest.raises(ValueError):                                  # beyond the 8192-token limit of the encoder path
        e.forward(np.zeros((1, 8200), np.int64), np.ones((1, 8200), np.int64))
    # a token id outside the embedding table (another model's tokenizer, -1 as padding) is an error, not an out-of-bounds read on the GPU
    ok_ids = np.full((2, 32), 7, np.int64)
    for bad_id in (cfg["vocab"], -1, 1 << 30):
        bad_ids = ok_ids.copy(); bad_ids[1, 5] = bad_id
        with pytest.raises(RuntimeError, match="outside the vocabulary"):
            e.forward(bad_ids, np.ones_like(ok_ids))
    with pytest.raises(RuntimeError, match="token type id"):
        e.forward(ok_ids, np.ones_like(ok_ids), np.full_like(ok_ids, cfg["type_vocab"]))
    assert np.isfinite(e.forward(ok_ids, np.ones_like(ok_ids))).all()
    e.close()
    dm = _hf_qwen3(128, 1, 2, 2, 64, 256, vocab=300)
    dec = vf.HipDecoder.from_hf(dm, pooling=2, normalize=False)
    bad_ids = np.full((1, 32), 5, np.int64); bad_ids[0, 31] = 300
    with pytest.raises(RuntimeError, match="outside the vocabulary"):
        dec.forward(bad_ids, np.ones_like(bad_ids))
    assert np.isfinite(dec.forward(np.full((1, 32), 5, np.int64), np.ones((1, 32), np.int64))).all()
    dec.close()


def test_pipeline_embed_retrieve_rerank_rank_chunk(vf):
    """BASELINE configs[3] in miniature, every stage on the GPU: embed the corpus (embed loop,
    src/load_data.py:120-128) -> FaissRetriever top-100 -> cross-encoder scores -> rank_chunk
    (src/utils/vllmManager.py:430-483) -> top bundles; rank_chunk is checked against the oracle's restatement fed
    with the same model outputs."""
    from datetime import datetime
    from oracle import ref_numpy as R

    class Tok:
        def __call__(self, a, b=None, padding=True, truncation=True, max_length=64, return_tensors="np"):
            a = [a] if isinstance(a, str) else list(a)
            b = [None] * len(a) if b is None else list(b)
            rows = []
            for x, y in zip(a, b):
                toks = [2] + [5 + (sum(map(ord, w)) * 31 % 900) for w in x.split()] + [3]
                if y is not None:
                    toks += [5 + (sum(map(ord, w)) * 31 % 900) for w in y.split()] + [3]
                rows.append(toks[:max_length])
            t = max(len(r) for r in rows)
            ids = np.ones((len(rows), t), np.int64)
            mask = np.zeros((len(rows), t), np.int64)
            for i, r in enumerate(rows):
                ids[i, :len(r)] = r
                mask[i, :len(r)] = 1
            return {"input_ids": ids, "attention_mask": mask}

    emb = vf.HipEmbeddings(Tok(), vf.HipEncoder.from_hf(_hf_bert(128, 2, 2, 512)), max_length=64, batch_size=100)
    rr = vf.HipReranker(Tok(), vf.HipEncoder.from_hf(_hf_xlmr_cls(128, 2, 2, 512)), max_length=64)
    docs = [f"filing {i} reports revenue item {i % 13} for segment {i % 5} in year {2015 + i % 9}" for i in range(1200)]
    docs[700] = docs[3]  # a verbatim duplicate: the 0.9 similarity rule must drop one of them
    vecs = emb.embed_documents(docs)                       # batches of 100, as load_data.py:151
    fr = vf.FaissRetriever(vecs, emb)
    question = "revenue item 3 for segment 3 in year 2018"
    I, D = fr.invoke([question], 100)
    assert I.shape == (1, 100) and np.all(np.diff(D[0]) <= 0)
    hits = [int(i) for i in I[0][:40]]
    if 3 in hits and 700 not in hits:
        hits[-1] = 700
    chunks = [{"page_content": docs[i], "bundle_id": j // 2, "metadata": {"date_published": f"20{15 + i % 9:02d}-0{1 + i % 9}-15"}}
              for j, i in enumerate(hits)]
    qt = datetime(2020, 6, 1)
    got = vf.rank_chunk(chunks, question, qt, rr, emb, chunk_topk=20, similar_threshhold=0.9)
    # restatement fed with the SAME model outputs (scores, embeddings): isolates the fusion / selection logic
    scores = rr.compute_score([[question, c["page_content"]] for c in chunks], batch_size=8)
    ts = vf.time_scores(qt, [c["metadata"]["date_published"] for c in chunks])
    embs = np.asarray(emb.embed_documents([c["page_content"] for c in chunks]), np.float32)
    want = R.rank_chunk([c["bundle_id"] for c in chunks], scores, ts, embs, 20, 0.9)
    assert got == want and 0 < len(got) <= 10 and len(set(got)) == len(got)


@pytest.mark.parametrize("M,N,K,epi,want_S", [
    (25600, 768, 768, 0, 4),       # 300 tiles: 44 in the partial round, cut in 4 (12 K-tiles -> 3 each)
    (51200, 768, 3072, 2, 2),      # the FFN-down product of 100 x 512 tokens: 88 tail tiles, cut in 2
    (51200, 2304, 768, 0, 4),      # the QKV product: 8 tail tiles
    (25600, 768, 320, 1, 2),       # 5 K-tiles: slices of 2 and 3 K-tiles, GELU epilogue
    (6656, 768, 3072, 2, 3),       # less than one round (13 x 512 tokens, 78 tiles): EVERY tile is cut, in 3; the sliced grid is padded to 80 tiles
    (3328, 1024, 1024, 0, 4),      # 52 tiles (padded to 56), cut in 4
    (6656, 1024, 4096, 2, 2),      # 104 tiles, K = 4096: the shape the dispatch cuts whole BY DEFAULT (FFN-down of 13 x 512 tokens, 1024 wide)
])
def test_gemm_splitk_tail_matches_torch_and_is_deterministic(vf, M, N, K, epi, want_S):
    """The tiles of the 8-phase kernel's partial last round are cut along K (one workgroup per slice; every slice hands the 16-row
    blocks it does not own over as fp32 partials through write-through stores, waits for the others and finishes ITS blocks, adding
    the partials in slice order): against torch fp32, against the same product without the split, and twice -- the result must
    not depend on the order in which the slices arrive.  A long-K product of less than half a round takes this form by default."""
    import ctypes
    import torch
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    L.vf_debug_gemm.restype = ctypes.c_int
    L.vf_debug_gemm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    L.vf_debug_splitk_tail.restype = ctypes.c_int
    L.vf_debug_splitk_tail.argtypes = [ctypes.c_int]
    dev = torch.device("cuda:0")
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    tiles = (M // 256) * (N // 256)
    ntail = tiles % cus if tiles > cus else tiles
    assert ntail > 0 and min(cus // ((ntail + 7) // 8 * 8), (K // 64) // 2, 4) == want_S
    g = torch.Generator(device=dev).manual_seed(7 * epi + K)
    A = (torch.randn(M, K, device=dev, generator=g) * 0.5).half()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.05).half()
    bias = torch.randn(N, device=dev, generator=g) * 0.1
    R = (torch.randn(M, N, device=dev, generator=g)).half()

    def run(kind=7):
        C = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
        rc = L.vf_debug_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr(), R.data_ptr(), C.data_ptr(), M, N, K, epi,
                             torch.cuda.current_stream().cuda_stream, kind)
        assert rc == 0
        torch.cuda.synchronize()
        return C
    L.vf_debug_splitk_stats.restype = ctypes.c_int
    L.vf_debug_splitk_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
    stats = (ctypes.c_uint * 2)()
    was = L.vf_debug_splitk_tail(2)                         # the general form (the default cuts long-K products only)
    try:
        run()                                               # (the first split launch allocates the workspace)
        L.vf_debug_splitk_stats(stats, -1)                  # reset the read-back counters
        c1, c2 = run(), run()
        assert L.vf_debug_splitk_stats(stats, -1) == 1
        L.vf_debug_splitk_tail(0)
        c0 = run()
    finally:
        L.vf_debug_splitk_tail(was)
    counted = (stats[0], stats[1])
    # wait bound zero: every slice but the last hands its whole partial over and leaves, the last arrival finishes the tile through
    # the take-over code (what a slice does after waiting 300 us for partners that cannot get a CU) -- same bits
    L.vf_debug_splitk_tail(2)
    try:
        L.vf_debug_splitk_stats(stats, 16)
        c3 = run()
    finally:
        L.vf_debug_splitk_stats(stats, 0)
        L.vf_debug_splitk_tail(was)
    assert torch.equal(c3, c1), "the take-over path of the split-K finish differs"
    if K >= 4096 and 2 * ((tiles + 7) // 8 * 8) <= cus:     # the dispatch's own choice for this shape is the same cut
        L.vf_debug_splitk_stats(stats, -1)
        c_auto = run(0)
        assert L.vf_debug_splitk_stats(stats, -1) == 1 and stats[0] + stats[1] == ntail
        assert torch.equal(c_auto, c1)
    # every cut tile is counted once per launch; the slices of a tile share an XCD (dispatch indices
    # congruent modulo 8), so the read-back normally goes through that XCD's L2 -- either way the result is the same
    print("split-K read-backs through L2 / from memory:", counted[0], counted[1])
    assert counted[0] + counted[1] == 2 * ntail
    assert torch.equal(c1, c2), "the split-K tail is not deterministic"
    ref = A[-2048:].float() @ W.float().T + bias          # the tail tiles are the LAST dispatch indices: their rows are checked ...
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 2:
        ref = ref + R[-2048:].float()
    assert not torch.isnan(c1).any()
    err = float((c1[-2048:].float() - ref).abs().max())
    d01 = float((c1.float() - c0.float()).abs().max())     # ... and the whole output against the unsplit launch
    print("split-K tail", (M, N, K, epi), "max err vs torch", err, "vs the unsplit launch", d01)
    assert err < 2e-2 and d01 < 1.6e-2      # fp16 output of O(10) values: one ulp is 7.8e-3


@pytest.mark.parametrize("M,N,K,epi,want_S", [
    (6656, 768, 3072, 2, 3),     # FFN-down + residual of 13 pairs x 512 tokens, 768 wide: 78 tiles, three slices of 16 K-tiles
    (6656, 768, 3072, 0, 3),
    (6656, 3072, 3072, 1, 0),    # 312 tiles: more than the CUs hold -- not cut
    (4096, 768, 3072, 1, 4),     # 8 pairs: 48 tiles, four slices of 12 K-tiles, GELU epilogue
    (6656, 1024, 4096, 2, 2),    # 1024 wide: 104 tiles, two slices of 32 K-tiles
    (6656, 768, 768, 2, 0),      # short K: slices would be 4 K-tiles -- not cut
])
def test_gemm9_whole_product_split_matches_torch_and_is_deterministic(vf, M, N, K, epi, want_S):
    """Round 5: a product of less than a round of 256 x 256 tiles with a long K is cut whole along K INSIDE the persistent kernel
    (k_gemm9_tn<EPI, true>: one item per workgroup, sk_coop_finish) -- what one rank of an 8-GPU data-parallel re-rank runs for
    FFN-down (/root/reference/src/utils/vllmManager.py:450-452 is the call being split).  Against torch fp32 on the same fp16
    operands, against the uncut launch, twice for determinism, and through the take-over path (wait bound zero)."""
    import ctypes
    import torch
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    L.vf_debug_gemm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    L.vf_debug_gemm9_split_launches.restype = ctypes.c_longlong
    L.vf_debug_splitk_stats.restype = ctypes.c_int
    L.vf_debug_splitk_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda", 0)
    cus = torch.cuda.get_device_properties(0).multi_processor_count & ~7
    tiles, nkt = (M // 256) * (N // 256), K // 64
    pad = (tiles + 7) // 8 * 8
    expect = next((c for c in (4, 3, 2) if K >= 2048 and pad * c <= cus and nkt // c >= 8), 0)
    assert expect == want_S, (expect, want_S)
    g = torch.Generator(device=dev).manual_seed(11 * epi + K + M)
    A = (torch.randn(M, K, device=dev, generator=g) * 0.5).half()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.05).half()
    bias = torch.randn(N, device=dev, generator=g) * 0.1
    R = (torch.randn(M, N, device=dev, generator=g)).half()

    def run():
        C = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
        assert L.vf_debug_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr(), R.data_ptr(), C.data_ptr(), M, N, K, epi,
                               torch.cuda.current_stream().cuda_stream, 0) == 0
        torch.cuda.synchronize()
        return C

    stats = (ctypes.c_uint * 2)()
    n0 = L.vf_debug_gemm9_split_launches()
    c1, c2 = run(), run()
    cut = L.vf_debug_gemm9_split_launches() - n0
    assert cut == (2 if want_S else 0), f"the dispatch cut {cut} of 2 launches (expected S = {want_S})"
    prev = L.vf_debug_gemm9_split(0)
    try:
        c0 = run()                                          # the same product uncut
    finally:
        L.vf_debug_gemm9_split(prev)
    assert L.vf_debug_gemm9_split_launches() - n0 == cut
    assert torch.equal(c1, c2), "the cut product is not deterministic"
    assert not torch.isnan(c1).any()
    ref = A.float() @ W.float().T + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 2:
        ref = ref + R.float()
    err, d01 = float((c1.float() - ref).abs().max()), float((c1.float() - c0.float()).abs().max())
    print("gemm9 whole-product cut", (M, N, K, epi), "S", want_S, "max err vs torch", err, "vs the uncut launch", d01)
    assert err < 2e-2 and d01 < 1.6e-2                      # fp16 output of O(10) values: one ulp is 7.8e-3
    if want_S:
        L.vf_debug_splitk_stats(stats, 16)                  # wait bound zero: every slice but the last hands over, the last finishes alone
        try:
            c3 = run()
        finally:
            L.vf_debug_splitk_stats(stats, 0)
        assert torch.equal(c3, c1), "the take-over path of the split-K finish differs"
        L.vf_debug_splitk_stats(stats, -1)
        run()
        assert L.vf_debug_splitk_stats(stats, -1) == 1 and stats[0] + stats[1] == tiles   # every tile finished once, counters back at zero


def test_two_handles_run_split_products_concurrently(vf):
    """Two encoder handles on two threads, each forward containing a product that is cut whole along K (13 x 512 tokens, 1024 wide:
    FFN-down = 104 tiles x 2 slices whose slices wait for one another): concurrent launches could each hold the CUs the other's
    missing slices need -- the bounded wait hands over and the results equal the sequential ones bit for bit."""
    import threading
    sys_path_tools = os.path.join(_ROOT, "tools")
    if sys_path_tools not in sys.path:
        sys.path.insert(0, sys_path_tools)
    from bench_rerank import random_encoder
    encs = []
    for seed in (0, 1):
        enc, _cfg = random_encoder("xlmr-large", head=1, seed=seed, vocab=2000)
        encs.append(enc)
    rng = np.random.default_rng(5)
    ids = rng.integers(5, 2000, size=(13, 512)).astype(np.int32)
    mask = np.ones_like(ids)
    want = [e.forward(ids, mask).copy() for e in encs]
    got = [[None] * 6, [None] * 6]
    def work(k):
        for it in range(6):
            got[k][it] = encs[k].forward(ids, mask).copy()
    ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ths: t.start()
    for t in ths: t.join(timeout=120)
    alive = [t.is_alive() for t in ths]
    for e in encs:
        if not any(alive):
            e.close()
    assert not any(alive), "concurrent split-K forwards did not finish"
    for k in range(2):
        for it in range(6):
            assert np.array_equal(got[k][it].view(np.uint32), want[k].view(np.uint32))


@pytest.mark.parametrize("kind", [3, 5, 7, 10])
@pytest.mark.parametrize("epi", [0, 1, 2, 11])
def test_gemm_kernels_match_torch(vf, kind, epi):
    """Every GEMM kernel (3 = register-staged 128x128, 5 = LDS-DMA 128x256 with two workgroups per CU, 7 = 8-phase 256x256, 10 = the
    persistent k_gemm9_tn) x every epilogue (bias, bias + erf-GELU, bias + residual, bias + quick-GELU) against torch fp32 on the same fp16 operands, at a shape with a
    ragged tile grid for the XCD-aware tile order (M = 1792 -> 14 / 7 m-tiles, N = 768, K = 320 -> 5 / 10 K-steps)."""
    import ctypes
    import torch
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    L.vf_debug_gemm.restype = ctypes.c_int
    L.vf_debug_gemm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(100 * kind + epi)
    shapes = [(1792, 768, 320), (512, 1024, 3072)]
    if kind == 10:
        # k_gemm9_tn (persistent, register-direct epilogue; K >= 256): fewer tiles than CUs, 320 / 384 tiles on 256 workgroups with odd
        # and even K-tile counts (LDS buffer parity alternates between a workgroup's tiles), an exact three rounds, a long K
        shapes = [(1792, 768, 320), (512, 1024, 3072), (10240, 2048, 320), (6144, 4096, 256), (16384, 3072, 448), (51200, 768, 768),
                  # long K with a partial last round: the remainder tiles are cut along K inside the kernel (600 tiles: 88 in two slices;
                  # 800 tiles: 32 in four; 288 tiles: 32 in four with 9 K-tiles per slice)
                  (51200, 768, 3072), (51200, 1024, 4096), (18432, 1024, 2304)]
    for (M, N, K) in shapes:
        if kind in (7, 10) and (M % 256 or N % 256):   # the 8-phase and the persistent kernel take 256 x 256 tiles only
            M, N = (M + 255) // 256 * 256, (N + 255) // 256 * 256
        A = (torch.randn(M, K, device=dev, generator=g) * 0.5).half()
        W = (torch.randn(N, K, device=dev, generator=g) * 0.05).half()
        bias = torch.randn(N, device=dev, generator=g)
        R = torch.randn(M, N, device=dev, generator=g).half()
        C = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
        rc = L.vf_debug_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr(), R.data_ptr(), C.data_ptr(), M, N, K, epi,
                             torch.cuda.current_stream().cuda_stream, kind)
        assert rc == 0
        torch.cuda.synchronize()
        ref = A.float() @ W.float().T + bias
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
        if epi == 11:                                  # quick-GELU, the vision tower's activation
            ref = ref * torch.sigmoid(1.702 * ref)
        del_big = M * N > 8_000_000
        if epi == 2:
            ref = ref.half().float() + R.float()   # the kernels round the biased product to fp16 before adding the residual
        err = (C.float() - ref).abs().max().item()
        assert not torch.isnan(C).any() and err < 2e-2 * max(1.0, (K / 3072) ** 0.5), (M, N, K, err)
        if kind == 10 and K >= 2048:       # the K-cut remainder: partials are summed in slice order, whoever arrives last
            C2 = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
            assert L.vf_debug_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr(), R.data_ptr(), C2.data_ptr(), M, N, K, epi,
                                   torch.cuda.current_stream().cuda_stream, kind) == 0
            torch.cuda.synchronize()
            assert torch.equal(C, C2), (M, N, K, "not deterministic")
        if del_big:
            del A, W, R, C, ref
            torch.cuda.empty_cache()


def test_encoder_handle_is_thread_safe(vf):
    """The reference shares one embedder between request threads without a lock (ragManager.py:17-30 singleton): concurrent
    forwards on one handle, with different batch shapes and per-call poolings, must each equal the serial result."""
    import threading
    enc = vf.HipEncoder.from_hf(_hf_bert(128, 2, 2, 512), pooling=0, normalize=True)
    rng = np.random.default_rng(8)
    jobs = []
    for i in range(12):
        b, t = int(rng.integers(1, 9)), int(rng.choice([16, 40, 64, 100]))
        ids, mask = _batch(rng, b, t, 1000)
        jobs.append((ids, mask, [None, 1, 2][i % 3]))
    want = [enc.forward(i, m, pooling=p, normalize=None if p is None else False) for i, m, p in jobs]
    got = [None] * len(jobs)
    def work(lo):
        for j in range(lo, len(jobs), 4):
            i, m, p = jobs[j]
            got[j] = enc.forward(i, m, pooling=p, normalize=None if p is None else False)
    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for th in threads: th.start()
    for th in threads: th.join()
    enc.close()
    for w, g in zip(want, got):
        assert np.array_equal(w.view(np.uint32), g.view(np.uint32))


# ---- decoder-only family: Qwen3-style embedder with last_token_pool, token-logit scorer -----------
def _hf_qwen3(hidden, layers, heads, kv_heads, head_dim, ffn, vocab=800, seed=3, causal_lm=False):
    import torch
    from transformers import Qwen3Config, Qwen3ForCausalLM, Qwen3Model
    torch.manual_seed(seed)
    cfg = Qwen3Config(vocab_size=vocab, hidden_size=hidden, intermediate_size=ffn, num_hidden_layers=layers,
                      num_attention_heads=heads, num_key_value_heads=kv_heads, head_dim=head_dim, max_position_embeddings=512,
                      rope_theta=1000000.0, tie_word_embeddings=False)
    m = (Qwen3ForCausalLM if causal_lm else Qwen3Model)(cfg).eval()
    with torch.no_grad():
        for p_ in m.parameters():                     # weights exactly representable in fp16 on both sides
            p_.copy_(p_.half().float())
            if p_.dim() == 1:
                p_.add_(torch.randn_like(p_) * 0.1).copy_(p_.half().float())   # norm gains away from 1
    return m


def _measured(test, **vals):
    """Every decoder test prints what it measured and appends it to gpurun_out/decoder_errors.jsonl; the asserted
    bounds are ~3x the values recorded in profiles/r03_decoder_errors.jsonl."""
    import json
    rec = {"test": test, **{k: (float(v) if np.ndim(v) == 0 else np.asarray(v).tolist()) for k, v in vals.items()}}
    print("measured", json.dumps(rec))
    try:
        os.makedirs(os.path.join(_ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(_ROOT, "gpurun_out", "decoder_errors.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass


def _embedding_errors(got, want):
    cos = (got * want).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))
    rel = np.abs(got - want).max() / np.abs(want).max()
    return float(1.0 - cos.min()), float(rel)


@pytest.mark.parametrize("name,hidden,layers,heads,kv_heads,head_dim,ffn,b,t,left_pad", [
    ("gqa-dh64", 256, 2, 4, 2, 64, 512, 3, 48, False),
    ("gqa-dh128-leftpad", 512, 3, 4, 1, 128, 1024, 4, 100, True),
    ("mha-dh128-long", 256, 2, 2, 2, 128, 768, 2, 300, False),
    ("gqa-dh128-2000", 256, 2, 4, 2, 128, 512, 1, 2000, False),   # beyond 512 tokens: RoPE table, 32 key tiles
    ("gqa-dh128-4096", 256, 2, 4, 2, 128, 512, 2, 4096, True),    # the reference's truncation length (step3_mul.py:200)
    ("gqa-dh64-3000", 256, 2, 4, 2, 64, 512, 1, 3000, False),     # a 3000-token chunk in the reference's own call shape
])
def test_decoder_embedder_matches_hf_fp32(vf, name, hidden, layers, heads, kv_heads, head_dim, ffn, b, t, left_pad):
    """last_token_pool embeddings of a random Qwen3-architecture model against HF fp32 on the CPU (same weights,
    fp16-rounded): RMSNorm, q/k-norm, RoPE, causal grouped-query attention (streaming kernel), SwiGLU, final norm,
    both padding sides (the reference's tokenizer is padding_side='left', continuous_retrieval.py:58)."""
    import torch
    from veritasfi_amd.retrieval import last_token_pool
    model = _hf_qwen3(hidden, layers, heads, kv_heads, head_dim, ffn)
    rng = np.random.default_rng(12)
    ids = rng.integers(5, 800, size=(b, t)).astype(np.int64)
    mask = np.ones((b, t), np.int64)
    for i in range(1, b):                                  # row 0 stays full
        n_pad = int(rng.integers(1, t // 2))
        if left_pad:
            mask[i, :n_pad] = 0
        else:
            mask[i, t - n_pad:] = 0
    with torch.no_grad():
        hs = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).last_hidden_state
        want = last_token_pool(hs, torch.from_numpy(mask)).numpy()
    dec = vf.HipDecoder.from_hf(model, pooling=2, normalize=False)
    got = dec.forward(ids, mask)
    dec.close()
    assert got.shape == (b, hidden)
    one_minus_cos, rel = _embedding_errors(got, want)
    _measured(f"decoder_embedder[{name}]", one_minus_cos=one_minus_cos, rel=rel)
    assert one_minus_cos < DEC_COS_TOL and rel < DEC_REL_TOL, (name, one_minus_cos, rel)


def test_decoder_token_logit_scorer_matches_hf_fp32(vf):
    """The LLM re-ranker's score = the logit of one token ("Yes") at the last position (stress_test.py:197,212-225)."""
    import torch
    model = _hf_qwen3(256, 2, 4, 2, 64, 512, causal_lm=True)
    rng = np.random.default_rng(13)
    ids = rng.integers(5, 800, size=(5, 64)).astype(np.int64)
    mask = np.ones((5, 64), np.int64)
    mask[2, :20] = 0                                       # left padding, as the reference pads re-ranker inputs
    yes = 123
    with torch.no_grad():
        logits = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).logits
        want = logits[:, -1, yes].numpy()
    dec = vf.HipDecoder.from_hf(model, score_token=yes)
    got = dec.forward(ids, mask)
    dec.close()
    err = float(np.abs(got - want).max())
    _measured("decoder_token_logit", abs_err=err, logit_scale=np.abs(want).max())
    assert got.shape == (5,) and err < DEC_LOGIT_TOL, (got, want)


class _StubLLMTokenizer:
    """Word-hash tokenizer with the HF methods the reference's get_inputs calls (left padding, as decoder re-rankers use)."""
    bos_token_id, pad_token_id, padding_side = 2, 0, "left"

    def __call__(self, text, return_tensors=None, add_special_tokens=False, max_length=None, truncation=False, **_):
        ids = [5 + (sum(map(ord, w)) * 31 + len(w)) % 780 for w in text.replace("\n", " \n ").split(" ") if w != ""]
        if text == "\n":
            ids = [4]
        if truncation and max_length is not None:
            ids = ids[:max_length]
        return {"input_ids": ids}

    def prepare_for_model(self, ids, pair_ids, truncation=None, max_length=None, **_):
        assert truncation == "only_second"
        over = len(ids) + len(pair_ids) - max_length
        if over > 0:
            pair_ids = pair_ids[:max(len(pair_ids) - over, 0)]
        return {"input_ids": list(ids) + list(pair_ids)}

    def pad(self, inputs, padding=True, max_length=None, pad_to_multiple_of=None, return_tensors=None):
        width = max(len(x["input_ids"]) for x in inputs)
        if pad_to_multiple_of:
            width = -(-width // pad_to_multiple_of) * pad_to_multiple_of
        ids = np.full((len(inputs), width), self.pad_token_id, np.int64)
        mask = np.zeros((len(inputs), width), np.int64)
        for i, x in enumerate(inputs):
            n = len(x["input_ids"])
            ids[i, width - n:], mask[i, width - n:] = x["input_ids"], 1
        return {"input_ids": ids, "attention_mask": mask}


def test_llm_reranker_compute_score(vf):
    """HipLLMReranker.compute_score == HF logits[:, -1, yes] on inputs built by the literal restatement of the reference's
    get_inputs (oracle/ref_rerank_inputs.py); the product's own input builder yields the same token rows, including the
    query (3/4) and only-second truncations."""
    import torch
    from oracle import ref_rerank_inputs as RI
    tok = _StubLLMTokenizer()
    rng = np.random.default_rng(14)
    words = ["revenue", "lotus", "margin", "battery", "delivery", "2023", "guidance", "segment", "cash", "vehicle"]
    sent = lambda n: " ".join(words[int(i)] for i in rng.integers(0, len(words), n))
    pairs = [[sent(int(rng.integers(3, 12))), sent(int(rng.integers(5, 60)))] for _ in range(11)]
    pairs.append([sent(80), sent(90)])                                 # forces both truncations at max_length = 96
    max_length = 96
    ref = RI.get_inputs(pairs, tok, max_length=max_length)
    mine = vf.build_llm_reranker_inputs(pairs, tok, max_length=max_length)
    for i, row in enumerate(mine):
        n = int(ref["attention_mask"][i].sum())
        assert row == ref["input_ids"][i][-n:].tolist()
    assert max(len(r) for r in mine) > max_length                      # pair clipped to max_length, then sep + prompt appended
    model = _hf_qwen3(256, 2, 4, 2, 64, 512, causal_lm=True)
    yes = tok("Yes")["input_ids"][0]
    with torch.no_grad():
        want = model(input_ids=torch.from_numpy(ref["input_ids"]), attention_mask=torch.from_numpy(ref["attention_mask"])).logits[:, -1, yes].numpy()
    rr = vf.HipLLMReranker(tok, vf.HipDecoder.from_hf(model, score_token=yes), max_length=max_length)
    s8 = rr.compute_score(pairs, batch_size=8)
    rr.fuse_batches = False
    s4 = rr.compute_score(pairs, batch_size=4)                         # literal micro-batching: same scores
    assert len(s8) == len(pairs) and all(isinstance(v, float) for v in s8)
    e8, e4 = float(np.abs(np.asarray(s8) - want).max()), float(np.abs(np.asarray(s4) - want).max())
    _measured("llm_reranker_compute_score", abs_err_fused=e8, abs_err_micro=e4, fused_vs_micro=np.abs(np.asarray(s8) - np.asarray(s4)).max(),
              logit_scale=np.abs(want).max())
    assert e8 < DEC_LOGIT_TOL and e4 < DEC_LOGIT_TOL
    assert np.abs(np.asarray(s8) - np.asarray(s4)).max() < DEC_LOGIT_TOL
    rr.decoder.close()


def _hf_gemma(hidden, layers, heads, kv_heads, head_dim, ffn, vocab=800, seed=5, causal_lm=False):
    import torch
    from transformers import GemmaConfig, GemmaForCausalLM, GemmaModel
    torch.manual_seed(seed)
    cfg = GemmaConfig(vocab_size=vocab, hidden_size=hidden, intermediate_size=ffn, num_hidden_layers=layers,
                      num_attention_heads=heads, num_key_value_heads=kv_heads, head_dim=head_dim, max_position_embeddings=2048,
                      rope_theta=10000.0)
    m = (GemmaForCausalLM if causal_lm else GemmaModel)(cfg).eval()
    with torch.no_grad():
        for p_ in m.parameters():
            if p_.dim() == 1:
                p_.add_(torch.randn_like(p_) * 0.1)          # zero-centred gains away from 0
            p_.copy_(p_.half().float())
    return m


@pytest.mark.parametrize("b,t,left_pad", [(3, 100, True), (2, 300, False), (2, 1100, True), (1, 4096, False)])
def test_gemma_style_decoder_matches_hf_fp32(vf, b, t, left_pad):
    """The configured re-ranker's architecture (config/example.yaml:9, bge-reranker-v2-gemma = gemma): head dim 256
    (Q tile in LDS, 32-key tiles), multi-query attention, (1 + w) RMSNorm, embeddings x sqrt(hidden), tanh-GELU gate;
    last-token hidden state and the logit of one token at the last position against HF fp32."""
    import torch
    from veritasfi_amd.retrieval import last_token_pool
    model = _hf_gemma(256, 2, 2, 1, 256, 512, causal_lm=True)
    rng = np.random.default_rng(15)
    ids = rng.integers(5, 800, size=(b, t)).astype(np.int64)
    mask = np.ones((b, t), np.int64)
    for i in range(1, b):
        n_pad = int(rng.integers(1, t // 2))
        if left_pad:
            mask[i, :n_pad] = 0
        else:
            mask[i, t - n_pad:] = 0
    tid, tm = torch.from_numpy(ids), torch.from_numpy(mask)
    with torch.no_grad():
        out = model(input_ids=tid, attention_mask=tm, output_hidden_states=True)
        want_h = last_token_pool(out.hidden_states[-1], tm).numpy()
        want_logit = last_token_pool(out.logits, tm)[:, 77].numpy()
    emb = vf.HipDecoder.from_hf(model, pooling=2, normalize=False)
    assert emb.cfg["head_dim"] == 256 and emb.cfg["act"] == 1 and emb.cfg["norm_plus_one"] == 1 and emb.cfg["qk_norm"] == 0
    got_h = emb.forward(ids, mask)
    emb.close()
    one_minus_cos, rel = _embedding_errors(got_h, want_h)
    sc = vf.HipDecoder.from_hf(model, score_token=77)
    got_logit = sc.forward(ids, mask)
    sc.close()
    lerr = float(np.abs(got_logit - want_logit).max())
    _measured(f"gemma_decoder[{b}x{t},left={left_pad}]", one_minus_cos=one_minus_cos, rel=rel, logit_abs_err=lerr,
              logit_scale=np.abs(want_logit).max())
    assert one_minus_cos < DEC_COS_TOL and rel < DEC_REL_TOL, (one_minus_cos, rel)
    assert lerr < DEC_LOGIT_TOL, (got_logit, want_logit)


# ---- FULL-DEPTH parity of the reference's configured decoder models (round-3 review, item 3): the real layer counts and
#      widths, vocabulary cut to 4096, against HF fp32 ON THE GPU (torch is plumbing in tests; the HF model is created there, so
#      neither 2 nor 3.6 billion fp32 parameters pass through host memory at once) ------------------------------------------
def _full_depth_rows(vf, n_rows=4, lo=64, hi=300, vocab=4096, seed=21):
    """Left-padded rows of lo..hi tokens built by the product's own input builder (the reference's get_inputs layout:
    bos + "A: " query + sep + "B: " passage + sep + prompt, experiments/profile/stress_test.py:97-146) on the stub tokenizer,
    token ids folded into the cut vocabulary."""
    tok = _StubLLMTokenizer()
    rng = np.random.default_rng(seed)
    words = ["revenue", "lotus", "margin", "battery", "delivery", "2023", "guidance", "segment", "cash", "vehicle", "emira", "eletre"]
    sent = lambda n: " ".join(words[int(i)] for i in rng.integers(0, len(words), n))
    targets = np.linspace(lo, hi, n_rows).astype(int)
    pairs = [[sent(8), sent(max(4, int(t) - 40))] for t in targets]
    rows = vf.build_llm_reranker_inputs(pairs, tok, max_length=1024)
    width = -(-max(len(r) for r in rows) // 32) * 32
    ids = np.zeros((len(rows), width), np.int64)
    mask = np.zeros((len(rows), width), np.int64)
    for i, r in enumerate(rows):
        ids[i, width - len(r):] = (np.asarray(r, np.int64) * 37 + 11) % (vocab - 8) + 5      # spread over the cut vocabulary
        mask[i, width - len(r):] = 1
    return ids, mask, tok


def _round_params_to_fp16(model, gain_noise):
    import torch
    with torch.no_grad():
        for p_ in model.parameters():
            if p_.dim() == 1 and gain_noise:
                p_.add_(torch.randn_like(p_) * gain_noise)
            p_.copy_(p_.half().float())


def test_full_depth_gemma_2b_geometry_yes_logit(vf):
    """bge-reranker-v2-gemma's geometry (config/example.yaml:9): 18 layers, hidden 2048, 8 query heads on ONE kv head of
    dim 256, GeGLU 16384, embeddings x sqrt(2048) -- the model bench.py times as `rerank_llm`.  The score the reference takes
    (logits[:, -1, yes], stress_test.py:197,212-225) and the last-token hidden state against HF fp32."""
    import torch
    from transformers import GemmaConfig, GemmaForCausalLM
    from veritasfi_amd.retrieval import last_token_pool
    ids, mask, tok = _full_depth_rows(vf)
    yes = int((tok("Yes")["input_ids"][0] * 37 + 11) % (4096 - 8) + 5)
    torch.manual_seed(31)
    cfg = GemmaConfig(vocab_size=4096, hidden_size=2048, intermediate_size=16384, num_hidden_layers=18, num_attention_heads=8,
                      num_key_value_heads=1, head_dim=256, max_position_embeddings=2048, rope_theta=10000.0)
    with torch.device("cuda"):
        model = GemmaForCausalLM(cfg).eval()
    _round_params_to_fp16(model, 0.1)
    tid, tm = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    with torch.no_grad():
        out = model(input_ids=tid, attention_mask=tm, output_hidden_states=True)
        want_h = last_token_pool(out.hidden_states[-1], tm).float().cpu().numpy()
        want_logit = out.logits[:, -1, yes].float().cpu().numpy()
    del out
    sc = vf.HipDecoder.from_hf(model, score_token=yes)
    got_logit = sc.forward(ids, mask)
    sc.close()
    emb = vf.HipDecoder.from_hf(model, pooling=2, normalize=False)
    got_h = emb.forward(ids, mask)
    emb.close()
    del model
    torch.cuda.empty_cache()
    one_minus_cos, rel = _embedding_errors(got_h, want_h)
    lerr = float(np.abs(got_logit - want_logit).max())
    _measured("full_depth_gemma_2b", one_minus_cos=one_minus_cos, rel=rel, logit_abs_err=lerr, logit_scale=np.abs(want_logit).max(),
              rows=ids.shape[0], width=ids.shape[1], layers=18)
    assert one_minus_cos < FULL_GEMMA_COS_TOL and rel < FULL_GEMMA_REL_TOL, (one_minus_cos, rel)
    assert lerr < FULL_GEMMA_LOGIT_TOL * max(1.0, float(np.abs(want_logit).max())), (got_logit, want_logit)


def test_full_depth_qwen3_embedding_4b_geometry_last_token(vf):
    """Qwen3-Embedding-4B's geometry (step3_mul.py:384, the reference's default embedder): 36 layers, hidden 2560, 32 query
    / 8 kv heads of dim 128, SwiGLU 9728, q/k-norm, RoPE theta 1e6.  last_token_pool embeddings (step3_mul.py:181-209)
    against HF fp32."""
    import torch
    from transformers import Qwen3Config, Qwen3Model
    from veritasfi_amd.retrieval import last_token_pool
    ids, mask, _ = _full_depth_rows(vf, seed=22)
    torch.manual_seed(32)
    cfg = Qwen3Config(vocab_size=4096, hidden_size=2560, intermediate_size=9728, num_hidden_layers=36, num_attention_heads=32,
                      num_key_value_heads=8, head_dim=128, max_position_embeddings=4096, rope_theta=1000000.0,
                      tie_word_embeddings=False)
    with torch.device("cuda"):
        model = Qwen3Model(cfg).eval()
    _round_params_to_fp16(model, 0.1)
    tid, tm = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
    with torch.no_grad():
        hs = model(input_ids=tid, attention_mask=tm).last_hidden_state
        want = last_token_pool(hs, tm).float().cpu().numpy()
    del hs
    dec = vf.HipDecoder.from_hf(model, pooling=2, normalize=False)
    got = dec.forward(ids, mask)
    dec.close()
    del model
    torch.cuda.empty_cache()
    one_minus_cos, rel = _embedding_errors(got, want)
    _measured("full_depth_qwen3_embedding_4b", one_minus_cos=one_minus_cos, rel=rel, rows=ids.shape[0], width=ids.shape[1], layers=36)
    assert got.shape == (ids.shape[0], 2560)
    assert one_minus_cos < FULL_QWEN_COS_TOL and rel < FULL_QWEN_REL_TOL, (one_minus_cos, rel)


def test_decoder_embedder_drop_in(vf):
    """HipDecoderEmbeddings (decoder model as `embedding_function`): batch-independent, unit-norm, and usable by the
    FaissRetriever exactly like the encoder embedder."""
    model = _hf_qwen3(256, 2, 4, 2, 64, 512)
    tok = _StubLLMTokenizer()
    emb = vf.HipDecoderEmbeddings(tok, vf.HipDecoder.from_hf(model, pooling=2, normalize=True), max_length=64, batch_size=7,
                                  query_instruction="query: ")
    docs = [f"filing {i} reports revenue item {i % 13} for segment {i % 5} in year {2015 + i % 9}" for i in range(60)]
    vecs = np.asarray(emb.embed_documents(docs), np.float32)
    assert vecs.shape == (60, 256) and np.abs(np.linalg.norm(vecs, axis=1) - 1.0).max() < 1e-3
    one = np.asarray(vf.HipDecoderEmbeddings(tok, emb.decoder, max_length=64, batch_size=1).embed_documents(docs[:5]), np.float32)
    assert np.abs(one - vecs[:5]).max() < 3e-3                     # batching / padding width do not change an embedding
    q = np.asarray(emb.embed_query(docs[17]), np.float32)
    assert q.shape == (256,) and not np.allclose(q, vecs[17], atol=1e-3)      # the instruction prefix is applied to queries
    fr = vf.FaissRetriever(vecs.tolist(), emb)
    I, D = fr.invoke([docs[3], docs[40]], 5)
    assert I.shape == (2, 5) and np.all(np.diff(D, axis=1) <= 0)
    emb.decoder.close()


# ---- the reference's CONFIGURED models: bge-m3 (XLM-R-large: hidden 1024, 16 heads, 24 layers, 8192 tokens;
#      config/example.yaml:3, src/utils/ragManager.py:50) and bge-reranker-large (BASELINE configs[4]) ---------------------
def _hf_xlmr_embedder(hidden, layers, heads, ffn, max_pos, vocab=1200, seed=7):
    import torch
    from transformers import XLMRobertaConfig, XLMRobertaModel
    torch.manual_seed(seed)
    cfg = XLMRobertaConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=ffn,
                           vocab_size=vocab, max_position_embeddings=max_pos, type_vocab_size=1, pad_token_id=1)
    return XLMRobertaModel(cfg, add_pooling_layer=False).eval().half().float()


def test_xlmr_large_shape_embedder_and_reranker(vf):
    """hidden 1024 / 16 heads / 24 layers / ffn 4096 -- the shape of bge-m3 and bge-reranker-large."""
    import torch
    rng = np.random.default_rng(11)
    ids, mask = _batch(rng, 3, 96, 1200, pad_id=1)
    emb = _hf_xlmr_embedder(1024, 24, 16, 4096, 514)
    with torch.no_grad():
        ref = emb(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).last_hidden_state[:, 0].numpy()
    ref = ref / np.linalg.norm(ref, axis=1, keepdims=True)
    enc = vf.HipEncoder.from_hf(emb, pooling=0, normalize=True)
    got = enc.forward(ids, mask)
    enc.close()
    cos, err = np.sum(got * ref, axis=1).min(), np.abs(got - ref).max()
    print("xlmr-large embedder: cos", cos, "max|d|", err)
    assert got.shape == (3, 1024) and cos > 0.99999 and err < 1.5e-3      # measured 3e-4 .. 5e-4 (24 post-LN layers in fp16)
    rr = _hf_xlmr_cls(1024, 24, 16, 4096)
    with torch.no_grad():
        want = rr(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).logits.view(-1).numpy()
    h = vf.HipEncoder.from_hf(rr)
    sc = h.forward(ids, mask)
    h.close()
    print("xlmr-large re-ranker logits", want, sc)
    assert np.abs(sc - want).max() < 3e-3 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("name,hidden,layers,heads,ffn,tol", [
    ("xlmr-base (bge-reranker-base, configs[3])", 768, 12, 12, 3072, 2.5e-3),     # measured 1.75e-3 of the logit range (6.4e-3 on a range of 3.67)
    ("xlmr-large (bge-reranker-large, configs[4])", 1024, 24, 16, 4096, 6.5e-3),   # measured 4.2e-3 of the range (6.9e-3 on 1.63; 24 post-LN layers in fp16)
])
def test_rerank_rank_order_at_the_configs_rerank_size(vf, name, hidden, layers, heads, ffn, tol):
    """100 pairs x 512 tokens -- what `compute_score` gets from rank_chunk (/root/reference/src/utils/vllmManager.py:450-452) in
    configs[3] / [4] -- at full depth against HF fp32: the score error stays inside the tolerance, the 100-item order is the
    reference's except between items closer than twice that error, and the 20 best (chunk_topk, what the LLM is shown) are the same
    set outside that band.  Ragged lengths, as real pairs are."""
    import time
    import torch
    model = _hf_xlmr_cls(hidden, layers, heads, ffn)
    with torch.no_grad():   # a random-init head gives logits within +-0.2: spread them to the range a trained re-ranker's logits have
        model.classifier.out_proj.weight.mul_(8.0)
        model.classifier.out_proj.weight.copy_(model.classifier.out_proj.weight.half().float())
    rng = np.random.default_rng(41)
    ids, mask = _batch(rng, 100, 512, 1200, pad_id=1)
    t0 = time.time()
    with torch.no_grad():
        ref = np.concatenate([model(input_ids=torch.from_numpy(ids[i:i + 20]), attention_mask=torch.from_numpy(mask[i:i + 20])).logits.view(-1).numpy()
                              for i in range(0, 100, 20)])
    t_ref = time.time() - t0
    rr = vf.HipEncoder.from_hf(model)
    got = rr.forward(ids, mask)
    rr.close()
    e, gap, ndisc = _assert_rank_order(ref, got, top=20)
    spread = float(ref.max() - ref.min())
    print(f"{name}: 100 x 512, HF fp32 {t_ref:.0f} s; max |d logit| {e:.2e} (logit range {spread:.2f}, smallest gap {gap:.2e}), discordant pairs {ndisc} of 4950")
    # (the head's output weights were scaled 8 x to give the logits a trained re-ranker's spread, which scales the error with them:
    #  the bar is relative to the logit RANGE -- the unscaled head's error is the 7e-4 of test_reranker_matches_torch_fp32)
    assert got.shape == (100,) and e < tol * spread
    assert ndisc <= 45          # (measured 4 / 21 of 4950 -- pairs whose reference logits differ in the third or fourth digit; each is checked against 2 e above)


@pytest.mark.parametrize("b,t,hidden,layers,heads,ffn", [(2, 1024, 256, 2, 4, 1024), (1, 8192, 128, 2, 2, 512), (2, 2000, 1024, 2, 16, 4096)])
def test_long_sequences_take_the_streaming_attention(vf, b, t, hidden, layers, heads, ffn):
    """t > 512 (bge-m3 accepts 8192 tokens): K / V no longer fit LDS whole, the encoder switches to the streaming
    attention kernel.  Ragged masks, RoBERTa position ids up to the end of an 8194-entry table."""
    import torch
    rng = np.random.default_rng(12)
    ids, mask = _batch(rng, b, t, 1200, pad_id=1)
    m = _hf_xlmr_embedder(hidden, layers, heads, ffn, 8194)
    with torch.no_grad():
        ref_h = m(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).last_hidden_state.numpy()
    ref = ref_h[:, 0] / np.linalg.norm(ref_h[:, 0], axis=1, keepdims=True)
    enc = vf.HipEncoder.from_hf(m, pooling=0, normalize=True)
    got = enc.forward(ids, mask)
    hs = enc.hidden_states(ids, mask)
    enc.close()
    herr = np.abs(hs - ref_h)[mask.astype(bool)]
    print("long sequence", (b, t, hidden), "cos", np.sum(got * ref, axis=1).min(), "max|d emb|", np.abs(got - ref).max(),
          "hidden mean/max", herr.mean(), herr.max())
    assert np.sum(got * ref, axis=1).min() > 0.99999 and np.abs(got - ref).max() < 1.5e-3
    assert herr.mean() < 2e-3 and herr.max() < 4e-2
    with pytest.raises(RuntimeError, match="position table"):
        short = vf.HipEncoder.from_hf(_hf_xlmr_embedder(128, 1, 2, 512, 514), pooling=0, normalize=True)
        try:
            short.forward(np.ones((1, 1024), np.int64), np.ones((1, 1024), np.int64))
        finally:
            short.close()


# ---- decoder family: fp32 residual stream + exported hidden states (reference loads these models in the checkpoint's
#      wider dtype, step3_mul.py:62-64, and pools outputs.last_hidden_state itself, :203-207) -------------------------------
def test_decoder_hidden_states_and_generic_get_embeddings_route(vf):
    import torch
    from veritasfi_amd.retrieval import get_embeddings, last_token_pool
    model = _hf_qwen3(256, 2, 4, 2, 64, 512)
    rng = np.random.default_rng(31)
    b, t = 3, 80
    ids = rng.integers(5, 800, size=(b, t)).astype(np.int64)
    mask = np.ones((b, t), np.int64)
    mask[1, :17] = 0                                          # left padding, as the reference's tokenizer pads
    mask[2, :40] = 0
    with torch.no_grad():
        ref = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).last_hidden_state.numpy()
    dec = vf.HipDecoder.from_hf(model, pooling=2, normalize=False)
    hs = dec.hidden_states(ids, mask)
    assert hs.shape == (b, t, 256) and hs.dtype == np.float32
    err = np.abs(hs - ref)[mask.astype(bool)]
    print("decoder hidden states: mean/max err", err.mean(), err.max(), "ref absmax", np.abs(ref).max())
    assert err.mean() < 3e-3 and err.max() < 5e-2
    # HF-signature callable: the reference's own pooling code runs on the exported states and agrees with the GPU pooling
    m = vf.HipDecoderModel(dec)
    out = m(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask))
    pooled_host = last_token_pool(out.last_hidden_state, torch.from_numpy(mask)).numpy()
    pooled_gpu = dec.forward(ids, mask)
    assert np.abs(pooled_host - pooled_gpu).max() < 5e-3 * np.abs(pooled_gpu).max()

    class Tok:
        padding_side = "left"
        def __call__(self, texts, padding=True, truncation=True, return_tensors="pt", max_length=None):
            r = [int(x) for x in texts]
            return {"input_ids": torch.from_numpy(ids[r]), "attention_mask": torch.from_numpy(mask[r])}

    class HiddenOnly:                                           # hides .pooled: forces the generic hidden-state route
        def __init__(self, mm): self.mm = mm
        def __call__(self, **kw): return self.mm(**kw)
    slow = get_embeddings(["0", "1", "2"], HiddenOnly(m), Tok(), "cpu", batch_size=2, pooling="last_token")
    fast = get_embeddings(["0", "1", "2"], m, Tok(), "cpu", batch_size=2, pooling="last_token")
    assert np.abs(slow - fast).max() < 5e-3 * np.abs(fast).max()
    dec.close()


def test_decoder_residual_beyond_fp16_range(vf):
    """A model whose residual stream leaves the fp16 range (|x| > 65504): layer 0's down-projection is scaled up so that
    the MLP writes ~1e5 into the stream (the 'massive activation' pattern of real checkpoints).  With the stream kept
    in fp32 the embeddings still match HF fp32; an fp16 stream turns them into inf / NaN."""
    import torch
    from veritasfi_amd.retrieval import last_token_pool
    model = _hf_qwen3(256, 3, 4, 2, 64, 512)
    with torch.no_grad():
        for lin, f in ((model.layers[0].mlp.up_proj, 300.0), (model.layers[0].mlp.down_proj, 4000.0)):
            lin.weight.mul_(f)
            lin.weight.copy_(lin.weight.half().float())     # still fp16-representable (|w| < 400)
    rng = np.random.default_rng(32)
    ids = rng.integers(5, 800, size=(2, 64)).astype(np.int64)
    mask = np.ones((2, 64), np.int64)
    peak = []
    hook = model.layers[1].register_forward_pre_hook(lambda mod, args: peak.append(float(args[0].abs().max())))
    with torch.no_grad():
        hs = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).last_hidden_state
    hook.remove()
    want = last_token_pool(hs, torch.from_numpy(mask)).numpy()
    print("residual peak entering layer 1:", peak[0])
    assert peak[0] > 65504.0, "the construction must push the residual stream past the fp16 range"
    dec = vf.HipDecoder.from_hf(model, pooling=2, normalize=False)
    got = dec.forward(ids, mask)
    dec.close()
    assert np.isfinite(got).all()
    cos = (got * want).sum(1) / (np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1))
    rel = np.abs(got - want).max() / np.abs(want).max()
    print("beyond-fp16 residual: cos", cos, "rel", rel)
    assert cos.min() > 0.999 and rel < 3e-2


@pytest.mark.parametrize("b,t,heads,ragged,growing,scale", [
    (6, 512, 12, False, False, 1.0),     # the re-rank shape, full sequences
    (9, 512, 4, True, False, 1.0),       # ragged lengths, a sequence without valid keys, a single valid key
    (5, 480, 2, True, True, 1.0),        # odd number of 32-query blocks; key norms grow along the sequence
    (4, 256, 3, False, True, 3.0),       # scores far past the fp16 range of the first reference: the overflow path
    (7, 96, 2, True, False, 1.0),
    (3, 32, 1, False, False, 1.0),
    (300, 64, 2, True, False, 1.0),      # more pairs than resident workgroups: the persistent loop, single-chunk refills
    (70, 512, 4, True, True, 2.0),       # 280 pairs on 256 workgroups: next-pair prefetch with skipped chunks
    (150, 128, 12, True, False, 1.0),    # 1800 pairs, one chunk per pair: the refill-at-pair-end path, several rounds
    (110, 256, 6, True, True, 1.0),      # 660 pairs on 512 resident workgroups: two chunks (last chunk published on demand)
    (30, 384, 12, False, False, 1.0),    # 360 pairs, three chunks, six waves
])
def test_attention_kernels_match_fp32_softmax(vf, b, t, heads, ragged, growing, scale):
    """k_attention2 (persistent, LDS-DMA prefetch, lazy softmax reference) and the two older kernels against an fp32
    softmax in torch, through the vf_debug_attention hook.  Rows of padded queries are not compared (HF never reads them)."""
    import ctypes
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_attention import make_case, reference, run
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    L.vf_debug_attention.restype = ctypes.c_int
    L.vf_debug_attention.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    qkv, mask = make_case(b, t, heads, dev, seed=11, scale=scale, ragged=ragged, growing=growing)
    rows = min(b, 12)
    ref = reference(qkv, mask, b, t, heads, rows=rows)
    valid = mask.reshape(b, t)[:rows].reshape(-1).bool()
    vmax = float(qkv[:, 2 * heads * 64:].float().abs().max())
    kinds = (2, 4)   # k_attention2, the LDS-DMA streaming kernel
    for kind in kinds:
        ctx = torch.full((b * t, heads * 64), float("nan"), dtype=torch.float16, device=dev)
        run(L, kind, qkv, mask, b, t, heads, ctx)
        torch.cuda.synchronize()
        got = ctx[:rows * t].float()[valid]
        assert bool(torch.isfinite(got).all()), (kind, "non-finite output")
        err = float((got - ref[valid]).abs().max())
        # measured <= 2.5e-4 * max|v| on these cases (fp16 probabilities and outputs); 3x that
        assert err <= 7.5e-4 * vmax, (kind, err, vmax)
    # the last pairs too (the persistent kernel's later rounds), kind 2 against the streaming kernel
    c2 = torch.zeros(b * t, heads * 64, dtype=torch.float16, device=dev)
    c3 = torch.zeros_like(c2)
    run(L, 2, qkv, mask, b, t, heads, c2)
    run(L, 4, qkv, mask, b, t, heads, c3)
    torch.cuda.synchronize()
    allv = mask.bool()
    assert float((c2.float()[allv] - c3.float()[allv]).abs().max()) <= 1.5e-3 * vmax


@pytest.mark.parametrize("b,tmax,heads,seed", [
    (40, 512, 4, 1),      # 160 pairs, lengths 32..512: one to four chunks per pair, idle waves on short pairs
    (90, 512, 4, 2),      # 360 pairs on 256 workgroups: hand-over between pairs of different chunk counts
    (200, 256, 6, 3),     # two workgroups per CU, one or two chunks
    (64, 96, 2, 4),
])
def test_attention_packed_variable_length(vf, b, tmax, heads, seed):
    """k_attention2 on PACKED sequences (each sequence its own length, rounded up to 32 rows) against the fp32 softmax
    of every sequence on its own."""
    import ctypes
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    L.vf_debug_attention_packed.restype = ctypes.c_int
    L.vf_debug_attention_packed.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    lens = rng.integers(1, tmax + 1, size=b)
    lens[0] = tmax                       # the longest sequence sizes the workgroup
    lens[1] = 1
    l32 = np.maximum(32, -(-lens // 32) * 32)
    off = np.concatenate([[0], np.cumsum(l32)]).astype(np.int32)
    rows, H = int(off[-1]), heads * 64
    g = torch.Generator(device=dev).manual_seed(seed)
    qkv = torch.randn(rows, 3 * H, device=dev, generator=g)
    qkv[:, :H] *= 0.125 * 1.4426950408889634
    qkv = qkv.half()
    mask_np = np.zeros(rows, np.int32)
    for i in range(b):
        mask_np[off[i]:off[i] + lens[i]] = 1
    mask = torch.from_numpy(mask_np).to(dev)
    seq_off = torch.from_numpy(off).to(dev)
    ctx = torch.full((rows, H), float("nan"), dtype=torch.float16, device=dev)
    rc = L.vf_debug_attention_packed(qkv.data_ptr(), mask.data_ptr(), seq_off.data_ptr(), b, int(l32.max()), heads, ctx.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    worst = 0.0
    x = qkv.float()
    for i in list(range(min(b, 12))) + list(range(max(12, b - 12), b)):     # head and tail of the pair list
        o0, n = int(off[i]), int(lens[i])
        q = x[o0:o0 + n, :H].reshape(n, heads, 64).transpose(0, 1)
        k = x[o0:o0 + n, H:2 * H].reshape(n, heads, 64).transpose(0, 1)
        v = x[o0:o0 + n, 2 * H:].reshape(n, heads, 64).transpose(0, 1)
        p = torch.softmax(q @ k.transpose(-1, -2) * 0.6931471805599453, dim=-1)
        ref = (p @ v).transpose(0, 1).reshape(n, H)
        got = ctx[o0:o0 + n].float()
        assert bool(torch.isfinite(got).all())
        worst = max(worst, float((got - ref).abs().max()))
    vmax = float(x[:, 2 * H:].abs().max())
    assert worst <= 7.5e-4 * vmax, (worst, vmax)


@pytest.mark.parametrize("kind", ["bert-embedder", "xlmr-reranker"])
def test_ragged_batch_takes_the_packed_forward_and_matches_torch(vf, kind):
    """A right-padded ragged batch (24 sequences, 9..256 tokens) runs PACKED (every sequence keeps ceil32(length) rows:
    vf_debug_packed_forwards counts it) and still matches HF fp32 -- BERT positions (embedder, CLS + L2) and RoBERTa
    positions (re-ranker logits); a batch that is not ragged enough, and one that is left-padded, take the padded path."""
    import ctypes
    import torch
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    L.vf_debug_packed_forwards.restype = ctypes.c_longlong
    rng = np.random.default_rng(21)
    b, t = 24, 256
    lens = rng.integers(9, t + 1, size=b)
    lens[0], lens[1] = t, 9
    if kind == "bert-embedder":
        m = _hf_bert(256, 3, 4, 512)
        pad_id, lo = 0, 5
    else:
        m = _hf_xlmr_cls(256, 3, 4, 512)
        pad_id, lo = 1, 5
    ids = rng.integers(lo, 900, size=(b, t)).astype(np.int64)
    mask = (np.arange(t)[None, :] < lens[:, None]).astype(np.int64)
    ids[mask == 0] = pad_id
    # BERT: segment ids too (second half of every sequence is segment 1, as a (query, passage) pair would be)
    tt = ((np.arange(t)[None, :] >= (lens[:, None] // 2)) & (mask == 1)).astype(np.int64) if kind == "bert-embedder" else None
    with torch.no_grad():
        if kind == "bert-embedder":
            ref = m(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask),
                    token_type_ids=torch.from_numpy(tt)).last_hidden_state[:, 0]
            ref = torch.nn.functional.normalize(ref, dim=-1).numpy()
        else:
            ref = m(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).logits[:, 0].numpy()
    enc = vf.HipEncoder.from_hf(m)
    try:
        n0 = L.vf_debug_packed_forwards()
        got = enc.forward(ids.astype(np.int32), mask.astype(np.int32), None if tt is None else tt.astype(np.int32))
        assert L.vf_debug_packed_forwards() == n0 + 1, "the ragged batch did not take the packed path"
        tol = 8e-4 if kind == "bert-embedder" else 2.5e-3
        assert float(np.abs(got - ref).max()) < tol, float(np.abs(got - ref).max())
        # nearly full batch: padded path
        full = np.ones_like(mask)
        enc.forward(ids.astype(np.int32), full.astype(np.int32))
        assert L.vf_debug_packed_forwards() == n0 + 1
        # left-padded batch: padded path, same answer as HF
        lmask = mask[:, ::-1].copy()
        lids = np.where(lmask == 1, rng.integers(lo, 900, size=(b, t)), pad_id)
        got_l = enc.forward(lids.astype(np.int32), lmask.astype(np.int32))
        assert L.vf_debug_packed_forwards() == n0 + 1 and np.isfinite(got_l).all()
    finally:
        enc.close()


@pytest.mark.parametrize("name,hidden,layers,heads,kv_heads,head_dim,ffn,b,t,left_pad", [
    ("qwen-dh64-right", 256, 2, 4, 2, 64, 512, 12, 160, False),
    ("qwen-dh128-left", 256, 2, 4, 1, 128, 512, 10, 224, True),
    ("qwen-dh128-left-unaligned", 256, 2, 4, 1, 128, 512, 10, 200, True),   # width % 32 != 0: HipDecoder.forward adds alignment
])                                                                            # columns on the RIGHT of a left-padded batch
def test_decoder_ragged_batch_takes_the_packed_forward(vf, name, hidden, layers, heads, kv_heads, head_dim, ffn, b, t, left_pad):
    """A ragged batch padded on one side runs PACKED through the decoder (rows = sum of ceil32(length), each token keeping
    its original column as RoPE position) and gives last_token_pool's embeddings of HF fp32; the token-logit head likewise."""
    import ctypes
    import torch
    from veritasfi_amd import _ffi
    from veritasfi_amd.retrieval import last_token_pool
    L = _ffi.lib()
    L.vf_debug_packed_forwards.restype = ctypes.c_longlong
    model = _hf_qwen3(hidden, layers, heads, kv_heads, head_dim, ffn, causal_lm=True)
    rng = np.random.default_rng(31)
    ids = rng.integers(5, 800, size=(b, t)).astype(np.int64)
    mask = np.ones((b, t), np.int64)
    for i in range(1, b):                                  # row 0 stays full
        n_pad = int(rng.integers(t // 4, t - 4))
        if left_pad:
            mask[i, :n_pad] = 0
        else:
            mask[i, t - n_pad:] = 0
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), output_hidden_states=True)
        want = last_token_pool(out.hidden_states[-1], torch.from_numpy(mask)).numpy()
        lengths = mask.sum(1)
        last = (t - 1) * np.ones(b, np.int64) if left_pad else lengths - 1
        want_logit = out.logits[torch.arange(b), torch.from_numpy(last), 7].numpy()
    dec = vf.HipDecoder.from_hf(model, pooling=2, normalize=False)
    n0 = L.vf_debug_packed_forwards()
    got = dec.forward(ids, mask)
    assert L.vf_debug_packed_forwards() == n0 + 1, "the ragged batch did not take the packed path"
    dec.close()
    one_minus_cos, rel = _embedding_errors(got, want)
    scorer = vf.HipDecoder.from_hf(model, score_token=7)
    n1 = L.vf_debug_packed_forwards()
    got_logit = scorer.forward(ids, mask)
    assert L.vf_debug_packed_forwards() == n1 + 1
    scorer.close()
    lerr = float(np.abs(got_logit - want_logit).max())
    _measured(f"decoder_ragged_packed[{name}]", one_minus_cos=one_minus_cos, rel=rel, logit_abs_err=lerr, logit_scale=np.abs(want_logit).max())
    assert one_minus_cos < DEC_COS_TOL and rel < DEC_REL_TOL, (name, one_minus_cos, rel)
    assert lerr < DEC_LOGIT_TOL, (got_logit, want_logit)


@pytest.mark.parametrize("act", ["silu", "gelu_tanh"])
def test_gated_mlp_gemm_matches_torch(vf, act):
    """The decoder's fused gate / up product (k_gemm8p_tn's gated epilogue: act(A Wg^T) * (A Wu^T) in one launch) against torch
    fp32, at a ragged m-tile count and an odd K-tile count."""
    import ctypes
    import torch
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    L.vf_debug_gemm.restype = ctypes.c_int
    L.vf_debug_gemm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(77)
    M, F, K = 256 * 17, 3072, 320
    A = (torch.randn(M, K, device=dev, generator=g) * 0.5).half()
    W = (torch.randn(2 * F, K, device=dev, generator=g) * 0.08).half()
    C = torch.full((M, F), float("nan"), device=dev, dtype=torch.float16)
    rc = L.vf_debug_gemm(A.data_ptr(), W.data_ptr(), None, None, C.data_ptr(), M, 2 * F, K, 4 if act == "silu" else 5,
                         torch.cuda.current_stream().cuda_stream, 0)
    assert rc == 0
    torch.cuda.synchronize()
    gate = A.float() @ W[:F].float().T
    up = A.float() @ W[F:].float().T
    ref = (torch.nn.functional.silu(gate) if act == "silu" else torch.nn.functional.gelu(gate, approximate="tanh")) * up
    err = (C.float() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert not torch.isnan(C).any() and err < 2e-3 * max(1.0, scale), (err, scale)


@pytest.mark.parametrize("kind", [0, 5, 7])
def test_fp32_residual_gemm_matches_torch(vf, kind):
    """out(fp32) = R(fp32) + A . W^T (the decoder's residual products) through every kernel that has the epilogue: the 8-phase
    kernel (7; also what 0 picks at this size: operands swapped, 16-byte read-modify-writes from registers), the 128 x 256
    DMA kernel (5)."""
    import ctypes
    import torch
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    L.vf_debug_gemm.restype = ctypes.c_int
    L.vf_debug_gemm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5 + kind)
    M, N, K = 256 * 24, 4096, 320
    A = (torch.randn(M, K, device=dev, generator=g) * 0.5).half()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.08).half()
    R = torch.randn(M, N, device=dev, generator=g) * 300.0          # a residual stream well beyond fp16's precision at its scale
    C = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32)
    rc = L.vf_debug_gemm(A.data_ptr(), W.data_ptr(), None, R.data_ptr(), C.data_ptr(), M, N, K, 3, torch.cuda.current_stream().cuda_stream, kind)
    assert rc == 0
    torch.cuda.synchronize()
    ref = R + A.float() @ W.float().T
    err = (C - ref).abs().max().item()
    assert not torch.isnan(C).any() and err < 2e-3, err


def test_long_ragged_batch_packs_through_the_streaming_attention(vf):
    """bge-m3's regime: documents of very different lengths up to 1536 tokens in one batch.  Wider than 512 tokens the packed
    forward runs the streaming attention kernel over the row offsets; the embeddings still match HF fp32."""
    import ctypes
    import torch
    from transformers import XLMRobertaConfig, XLMRobertaModel
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    L.vf_debug_packed_forwards.restype = ctypes.c_longlong
    torch.manual_seed(4)
    cfg = XLMRobertaConfig(hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512, vocab_size=900,
                           max_position_embeddings=1600, type_vocab_size=1, pad_token_id=1)
    m = XLMRobertaModel(cfg, add_pooling_layer=False).eval()
    m = m.half().float()
    rng = np.random.default_rng(8)
    b, t = 5, 1536
    lens = np.array([1536, 700, 33, 1200, 260])
    ids = rng.integers(5, 900, size=(b, t)).astype(np.int64)
    mask = (np.arange(t)[None, :] < lens[:, None]).astype(np.int64)
    ids[mask == 0] = 1
    with torch.no_grad():
        ref = m(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).last_hidden_state[:, 0]
        ref = torch.nn.functional.normalize(ref, dim=-1).numpy()
    enc = vf.HipEncoder.from_hf(m)
    try:
        n0 = L.vf_debug_packed_forwards()
        got = enc.forward(ids.astype(np.int32), mask.astype(np.int32))
        assert L.vf_debug_packed_forwards() == n0 + 1
        assert float(np.abs(got - ref).max()) < 8e-4, float(np.abs(got - ref).max())
    finally:
        enc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,epi,kind", [
    (6656, 3072, 768, 1, 0),      # FFN-up of 13 pairs: 312 tiles = 1.2 rounds
    (12800, 3072, 768, 1, 0),     # FFN-up of 25 pairs: 600 tiles = 2.3 rounds
    (12800, 768, 3072, 2, 0),     # FFN-down of 25 pairs: 150 tiles, long K (three workgroups per tile)
    (25600, 768, 3072, 2, 0),     # FFN-down of 50 pairs
    (6656, 2304, 768, 0, 12),     # forced: 234 tiles (every workgroup's range inside one or two tiles)
    (6912, 768, 3072, 2, 12),     # forced: 81 tiles (not a multiple of 8: XCD chunks of 11 and 10), four workgroups per tile
    (2048, 1024, 1024, 0, 12),    # forced: 32 tiles, 16 K-tiles each: 2 K-tiles per workgroup -- below the gate, must fall back
])
def test_gemm9_stream_k_matches_torch_and_is_deterministic(vf, M, N, K, epi, kind):
    """Round 5: stream-K inside the persistent product kernel (k_gemm9_tn<EPI, 2>): the K-tiles of an XCD's tiles are dealt out evenly,
    a workgroup dumps the head of the tile it cannot complete first and finishes the tile begun below it last.  Against torch fp32 on
    the same fp16 operands, against the whole-tile launch, and twice for determinism (the partials are added in a fixed order)."""
    import ctypes
    import torch
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    L.vf_debug_gemm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int]
    L.vf_debug_gemm9_streamk_launches.restype = ctypes.c_longlong
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(7 * epi + K + M + N)
    A = (torch.randn(M, K, device=dev, generator=g) * 0.5).half()
    W = (torch.randn(N, K, device=dev, generator=g) * 0.05).half()
    bias = torch.randn(N, device=dev, generator=g) * 0.1
    R = (torch.randn(M, N, device=dev, generator=g)).half()

    def run(k):
        C = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
        rc = L.vf_debug_gemm(A.data_ptr(), W.data_ptr(), bias.data_ptr(), R.data_ptr(), C.data_ptr(), M, N, K, epi,
                             torch.cuda.current_stream().cuda_stream, k)
        assert rc == 0, rc
        torch.cuda.synchronize()
        return C

    n0 = L.vf_debug_gemm9_streamk_launches()
    was = L.vf_debug_gemm9_streamk(1)                      # (off by default: measured slower on every shape of the forward)
    try:
        c1, c2, c3 = run(kind), run(kind), run(kind)
    finally:
        L.vf_debug_gemm9_streamk(was)
    took = L.vf_debug_gemm9_streamk_launches() - n0
    below_gate = (M // 256) * (N // 256) // 8 * (K // 64) // 32 < 6
    assert took == (0 if below_gate else 3), f"{took} of 3 launches went stream-K"
    prev = L.vf_debug_gemm9_streamk(0)
    try:
        c0 = run(0)                                        # whole tiles
    finally:
        L.vf_debug_gemm9_streamk(prev)
    assert L.vf_debug_gemm9_streamk_launches() - n0 == took
    assert not torch.isnan(c1).any()
    assert torch.equal(c1, c2) and torch.equal(c1, c3), "stream-K is not deterministic"
    ref = A.float() @ W.float().T + bias
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 2:
        ref = ref + R.float()
    err, d01 = float((c1.float() - ref).abs().max()), float((c1.float() - c0.float()).abs().max())
    print("gemm9 stream-K", (M, N, K, epi, kind), "max err vs torch", err, "vs whole tiles", d01)
    assert err < 2e-2 and d01 < 1.6e-2


@pytest.mark.gpu
@pytest.mark.parametrize("b,t,heads,ragged", [
    (25, 512, 12, False),    # one rank's share at 4 GPUs: 300 pairs = 256 + 44 -> the last 44 as 88 half items
    (50, 512, 12, True),     # 600 = 2 x 256 + 88
    (6, 512, 12, True),      # 72 pairs, at most half the CUs: every pair as two half items
    (30, 384, 12, True),     # six waves: halves of three
    (13, 512, 12, False),    # 156 pairs: more than half the CUs, one round -- no half items
])
def test_attention_half_items_are_bit_equal_to_whole_pairs(vf, b, t, heads, ragged):
    """Round 5: the pairs of a partial last round (and small batches altogether) are walked by k_attention2 as two half items each, half the
    waves active in either.  A query's arithmetic does not depend on which workgroup runs it: bit-equal to the run with whole pairs."""
    import ctypes
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_attention import make_case, run
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    L.vf_debug_attention.restype = ctypes.c_int
    L.vf_debug_attention.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_int]
    dev = torch.device("cuda:0")
    qkv, mask = make_case(b, t, heads, dev, seed=5, scale=1.0, ragged=ragged, growing=False)
    outs = []
    for on in (1, 0, 1):
        was = L.vf_debug_attention_halves(on)
        try:
            ctx = torch.zeros(b * t, heads * 64, dtype=torch.float16, device=dev)
            run(L, 2, qkv, mask, b, t, heads, ctx)
            torch.cuda.synchronize()
        finally:
            L.vf_debug_attention_halves(was)
        outs.append(ctx)
    valid = mask.bool()
    assert bool(torch.isfinite(outs[0].float()[valid]).all())
    assert torch.equal(outs[0][valid], outs[1][valid]) and torch.equal(outs[0][valid], outs[2][valid])


def test_decoder_first_forward_is_right_while_another_handle_is_busy(vf):
    """The decoder's RoPE table is written when its workspace is (re)allocated -- on the handle's OWN stream since round 6 (it was launched on
    the legacy NULL stream, which a non-blocking stream does not order itself behind: the first forward after an allocation could read a
    table not written yet -- round-5 advisor).  Here the FIRST forward of fresh decoder handles runs while an encoder handle on another
    thread replays captured graphs back to back; every first result must equal the handle's later ones bit for bit, and a larger batch
    (a re-allocation: a new table) likewise."""
    import threading
    sys_path_tools = os.path.join(_ROOT, "tools")
    if sys_path_tools not in sys.path:
        sys.path.insert(0, sys_path_tools)
    from bench_rerank import random_encoder
    busy, _cfg = random_encoder("bert-base", head=0, vocab=1000)
    rng = np.random.default_rng(9)
    bids = rng.integers(5, 1000, size=(2, 64)).astype(np.int32)          # 128 rows: the graph-replay path
    stop = threading.Event()

    def hammer():
        while not stop.is_set():
            busy.forward(bids, np.ones_like(bids))

    th = threading.Thread(target=hammer)
    th.start()
    try:
        model = _hf_qwen3(128, 2, 2, 1, 64, 256)
        ids = rng.integers(5, 800, size=(6, 96)).astype(np.int32)
        mask = np.ones_like(ids)
        big_ids = rng.integers(5, 800, size=(24, 160)).astype(np.int32)
        for _ in range(4):
            dec = vf.HipDecoder.from_hf(model, pooling=2, normalize=False)
            first = dec.forward(ids, mask).copy()                         # first call: allocates the workspace, writes the table
            again = dec.forward(ids, mask)
            assert np.isfinite(first).all() and np.array_equal(first.view(np.uint32), again.view(np.uint32))
            grown = dec.forward(big_ids, np.ones_like(big_ids)).copy()    # a larger batch: re-allocation
            assert np.array_equal(grown.view(np.uint32), dec.forward(big_ids, np.ones_like(big_ids)).view(np.uint32))
            assert np.array_equal(first.view(np.uint32), dec.forward(ids, mask).view(np.uint32))
            dec.close()
    finally:
        stop.set()
        th.join(timeout=60)
        busy.close()
    assert not th.is_alive()
