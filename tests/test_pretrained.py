"""One-line construction (veritasfi_amd/pretrained.py) and REAL fast tokenizers driving the product.

CPU part: the tokenizers behave as their families do (pair templates, token types, truncation='only_second', padding side), the
sentence-transformers layout of a model directory is read as SentenceTransformer would assemble it, and from_config resolves the
reference's YAML keys (/root/reference/config/example.yaml:1-15) plus the two optional ones.  GPU part: HipEmbeddings / HipReranker /
HipLLMReranker built by from_pretrained from directories on disk, fed by those tokenizers, against HF fp32 models fed THE SAME
tokenizer output."""
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tokenizers_synth as TS  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEXTS = ["the revenue of w1 was w2 in the fiscal quarter", "cash flow guidance for w7 and w8", "w3",
         "deliveries by w9 were not in the table but in the figure " * 4, "margin"]
QUERY = "what was the revenue of w1 in the quarter"


def test_tokenizer_families_behave_as_the_reference_uses_them(tmp_path):
    bt, xt, gt = TS.bert_tokenizer(tmp_path), TS.xlmr_tokenizer(), TS.gemma_tokenizer()
    e = bt([QUERY] * 2, [TEXTS[0], TEXTS[3]], padding=True, truncation="only_second", max_length=24, return_tensors="np")
    assert e["input_ids"].shape == (2, 24) and set(np.unique(e["token_type_ids"])) == {0, 1}
    first_len = int((e["token_type_ids"][1] == 0).sum())
    assert (e["input_ids"][0, :first_len] == e["input_ids"][1, :first_len]).all()          # the query survives truncation whole
    assert e["input_ids"][1, -1] == bt.sep_token_id and e["attention_mask"][1].all()        # the passage was cut to fit
    assert e["attention_mask"][0].sum() < 24 and e["input_ids"][0, -1] == bt.pad_token_id   # right padding
    x = xt([QUERY], [TEXTS[0]], return_tensors="np")["input_ids"][0]
    seps = np.nonzero(x == 2)[0]
    assert x[0] == 0 and len(seps) == 3 and seps[1] == seps[0] + 1 and "token_type_ids" not in xt([QUERY], [TEXTS[0]])   # <s> A </s></s> B </s>
    assert xt.pad_token_id == 1
    g = gt([TEXTS[0], TEXTS[2]], padding=True, return_tensors="np")
    assert gt.padding_side == "left" and g["input_ids"][1, 0] == gt.pad_token_id and g["input_ids"][1, -2] == gt.bos_token_id
    assert gt("Yes", add_special_tokens=False)["input_ids"] == [4] and gt("\n", add_special_tokens=False)["input_ids"] == [6]


def test_sentence_transformers_layout_is_read_as_sentence_transformers_assembles_it(tmp_path):
    from veritasfi_amd.pretrained import read_sentence_transformers_layout
    bt = TS.bert_tokenizer(tmp_path)
    m = TS.tiny_bert(len(bt))
    d1 = TS.write_st_dir(str(tmp_path / "cls_norm"), bt, m, pooling="cls", normalize=True, max_seq_length=48)
    lay = read_sentence_transformers_layout(d1)
    assert lay["pooling"] == "cls" and lay["normalize"] is True and lay["max_seq_length"] == 48 and lay["layout"] == "modules.json"
    assert os.path.samefile(lay["transformer_dir"], d1)
    d2 = TS.write_st_dir(str(tmp_path / "mean"), bt, m, pooling="mean", normalize=False)
    assert read_sentence_transformers_layout(d2)["pooling"] == "mean" and read_sentence_transformers_layout(d2)["normalize"] is False
    d3 = TS.write_st_dir(str(tmp_path / "plain"), bt, m, modules=False)
    assert read_sentence_transformers_layout(d3) == dict(transformer_dir=d3, pooling="mean", normalize=False, max_seq_length=None, layout="plain")
    d4 = TS.write_st_dir(str(tmp_path / "last"), bt, m, pooling="lasttoken")
    assert read_sentence_transformers_layout(d4)["pooling"] == "lasttoken"
    mods = json.load(open(os.path.join(d1, "modules.json")))
    mods.append({"idx": 3, "name": "3", "path": "3_Dense", "type": "sentence_transformers.models.Dense"})
    json.dump(mods, open(os.path.join(d1, "modules.json"), "w"))
    with pytest.raises(ValueError, match="Dense"):
        read_sentence_transformers_layout(d1)            # a module without a HIP counterpart is refused by name, not skipped
    with pytest.raises(FileNotFoundError):
        from veritasfi_amd.pretrained import resolve_model_dir
        resolve_model_dir(str(tmp_path / "no_such_model"))


def test_from_config_reads_the_reference_keys_and_the_two_optional_ones(tmp_path):
    import yaml
    import veritasfi_amd as vf
    # (the keys and values of /root/reference/config/example.yaml:1-15; the reference tree is not read at test time)
    ref_cfg = {"persist_directory": "path/to/db", "embeddings_model_name": "BAAI/bge-m3", "llm_model_name": "Qwen/Qwen2___5-72B-Instruct-AWQ",
               "llm_base_url": "http://127.0.0.1:8000/v1", "llm_api_key": "EMPTY", "rerank_model": "BAAI/bge-reranker-v2-gemma", "rerank_topk": 5,
               "log_level": "INFO", "bearer_token": "your_bearer_token_here"}
    parts = vf.from_config(ref_cfg, load_models=False)                       # an existing YAML works unchanged
    assert parts.device_ids == [0] and parts.corpus_dtype == "f32" and parts.rerank_topk == ref_cfg.get("rerank_topk")
    assert parts.retriever_cls.func is vf.FaissRetriever and parts.retriever_cls.keywords == {"device_id": 0, "device_ids": None, "corpus_dtype": "f32"}
    p = tmp_path / "cfg.yaml"
    p.write_text(yaml.safe_dump(dict(ref_cfg, device_ids=[2, 3], corpus_dtype="FP8")))
    parts = vf.from_config(str(p), load_models=False)
    assert parts.device_ids == [2, 3] and parts.corpus_dtype == "fp8" and parts.retriever_cls.keywords["device_ids"] == [2, 3]
    with pytest.raises(ValueError, match="corpus_dtype"):
        vf.from_config(dict(ref_cfg, corpus_dtype="int4"), load_models=False)
    with pytest.raises(KeyError, match="rerank_model"):
        vf.from_config({"embeddings_model_name": "x"}, load_models=False)


# ---- GPU: the loaders build working objects, real tokenizers feed them -------------------------------------------------------------
def _hf_sentence_embeddings(model, tok, texts, pooling, normalize, max_length):
    import torch
    enc = tok(texts, padding=True, truncation=True, max_length=max_length, return_tensors="pt")
    with torch.no_grad():
        h = model(**enc).last_hidden_state
    m = enc["attention_mask"].unsqueeze(-1).float()
    if pooling == "cls":
        e = h[:, 0]
    elif pooling == "mean":
        e = (h * m).sum(1) / m.sum(1).clamp(min=1e-9)
    else:
        e = h[torch.arange(h.shape[0]), enc["attention_mask"].sum(1) - 1] if tok.padding_side == "right" else h[:, -1]
    if normalize:
        e = torch.nn.functional.normalize(e, dim=-1)
    return e.numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("pooling,normalize,modules", [("cls", True, True), ("mean", False, True), ("mean", False, False)])
def test_embedder_from_pretrained_matches_hf_fp32_on_the_same_tokenizer_output(tmp_path, pooling, normalize, modules):
    import veritasfi_amd as vf
    from transformers import AutoModel
    bt = TS.bert_tokenizer(tmp_path)
    d = TS.write_st_dir(str(tmp_path / "emb"), bt, TS.tiny_bert(len(bt)), pooling=pooling, normalize=normalize, max_seq_length=40, modules=modules)
    emb = vf.HuggingFaceEmbeddings(model_name=d)                            # the reference's constructor call (ragManager.py:50)
    try:
        assert isinstance(emb, vf.HipEmbeddings) and emb.layout["pooling"] == pooling
        assert emb.max_length == (40 if modules else 128)                   # max_seq_length of the layout, else what the position table holds
        want = _hf_sentence_embeddings(AutoModel.from_pretrained(d).eval(), bt, TEXTS, pooling, normalize, emb.max_length)
        got = np.asarray(emb.embed_documents(TEXTS), np.float32)
        err = float(np.abs(got - want).max())
        print("from_pretrained embedder", pooling, normalize, modules, "max |d|", err)
        assert got.shape == want.shape and err < (8e-4 if normalize else 6e-3)
        assert np.allclose(emb.embed_query(TEXTS[1]), got[1], atol=2e-3 if normalize else 1e-2)
    finally:
        emb.encoder.close()


@pytest.mark.gpu
def test_pair_inputs_with_token_types_match_hf(tmp_path):
    """BERT pair encoding: [CLS] A [SEP] B [SEP] with token_type_ids 0 / 1 and truncation='only_second' -- the hidden states of the
    HIP forward against HF fp32 on exactly that tokenizer output."""
    import torch
    import veritasfi_amd as vf
    bt = TS.bert_tokenizer(tmp_path)
    model = TS.tiny_bert(len(bt), type_vocab=2)
    enc = bt([QUERY] * len(TEXTS), TEXTS, padding=True, truncation="only_second", max_length=40, return_tensors="pt")
    assert enc["token_type_ids"].max() == 1
    with torch.no_grad():
        want = model(**enc).last_hidden_state.numpy()
    h = vf.HipEncoder.from_hf(model, pooling=0, normalize=False)
    try:
        got = h.hidden_states(enc["input_ids"].numpy(), enc["attention_mask"].numpy(), enc["token_type_ids"].numpy())
        no_types = h.hidden_states(enc["input_ids"].numpy(), enc["attention_mask"].numpy())
    finally:
        h.close()
    m = enc["attention_mask"].numpy().astype(bool)
    err = float(np.abs(got - want)[m].max())
    assert err < 2e-2 and float(np.abs(no_types - want)[m].max()) > 5 * err      # the token types matter and are honoured


@pytest.mark.gpu
def test_cross_encoder_from_pretrained_scores_real_pair_encodings(tmp_path):
    """FlagReranker-style scoring through HipReranker.from_pretrained: XLM-R pair template <s> A </s></s> B </s>, right padding with
    id 1, truncation at max_length -- logits and rank order against XLMRobertaForSequenceClassification fp32 on the same encodings."""
    import torch
    import veritasfi_amd as vf
    from transformers import AutoModelForSequenceClassification
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_encoder import _assert_rank_order
    xt = TS.xlmr_tokenizer()
    d = str(tmp_path / "rr")
    TS.tiny_xlmr_cross_encoder(len(xt)).save_pretrained(d)
    xt.save_pretrained(d)
    rr = vf.HipReranker.from_pretrained(d, max_length=48)
    try:
        assert isinstance(rr, vf.HipReranker)
        passages = TEXTS + [f"w{i} w{i + 1} revenue and margin of w{i + 2}" for i in range(20)]
        pairs = [[QUERY, p] for p in passages]
        got = np.asarray(rr.compute_score(pairs, batch_size=8), np.float64)
        hf = AutoModelForSequenceClassification.from_pretrained(d).eval()
        enc = xt([p[0] for p in pairs], [p[1] for p in pairs], padding=True, truncation=True, max_length=48, return_tensors="pt")
        assert enc["input_ids"].shape[1] == 48                                # the long passage was truncated
        with torch.no_grad():
            want = hf(**enc).logits.view(-1).numpy().astype(np.float64)
        e, gap, ndisc = _assert_rank_order(want, got, top=5)
        print("cross-encoder from_pretrained: max |d logit|", e, "smallest gap", gap, "discordant pairs", ndisc)
        assert e < 2.5e-3 * max(1.0, float(np.abs(want).max()))
    finally:
        rr.encoder.close()


@pytest.mark.gpu
def test_llm_reranker_from_pretrained_matches_the_reference_input_construction(tmp_path):
    """FlagLLMReranker(config['rerank_model'], devices='cuda', use_fp16=True) (vllmChatService.py:90) -> HipLLMReranker: a gemma
    checkpoint + a left-padding tokenizer on disk; compute_score against HF GemmaForCausalLM fp32 fed the inputs the reference's
    get_inputs builds (oracle/ref_rerank_inputs.py, pinned by fixture g7): logits[:, -1, yes_loc]."""
    import torch
    import veritasfi_amd as vf
    from transformers import AutoModelForCausalLM
    from oracle import ref_rerank_inputs as RI
    gt = TS.gemma_tokenizer()
    d = str(tmp_path / "llm")
    TS.tiny_gemma_lm(len(gt)).save_pretrained(d)
    gt.save_pretrained(d)
    rr = vf.FlagLLMReranker(d, devices="cuda", use_fp16=True, max_length=64)
    try:
        assert isinstance(rr, vf.HipLLMReranker) and rr.yes_loc == 4 and rr.pad_id == 0
        pairs = [[QUERY, p] for p in TEXTS]
        got = np.asarray(rr.compute_score(pairs, batch_size=8), np.float64)
        hf = AutoModelForCausalLM.from_pretrained(d).eval()
        # the inputs as FlagLLMReranker lays them out (bos + "A: q" | "\n" + "B: p" truncated only_second | "\n" + prompt), LEFT-padded to a
        # multiple of 8: build_llm_reranker_inputs is pinned to the reference's get_inputs by fixture g7 (tests/test_control_flow_golden.py;
        # this transformers release's fast tokenizers have no prepare_for_model, so get_inputs itself cannot run on one)
        rows = vf.build_llm_reranker_inputs(pairs, gt, max_length=64)
        width = -(-max(len(r) for r in rows) // 8) * 8
        ids = np.zeros((len(rows), width), np.int64)
        mask = np.zeros_like(ids)
        for j, r in enumerate(rows):
            ids[j, width - len(r):], mask[j, width - len(r):] = r, 1
        assert mask[2, 0] == 0 and ids[0, -1] != 0 and all(r[0] == gt.bos_token_id for r in rows)   # left padding; bos first
        assert max(len(r) for r in rows) <= 64 + len(gt("\n", add_special_tokens=False)["input_ids"]) + len(gt(RI.DEFAULT_PROMPT, add_special_tokens=False)["input_ids"])
        with torch.no_grad():
            logits = hf(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask)).logits
        want = logits[:, -1, 4].numpy().astype(np.float64)
        err = float(np.abs(got - want).max())
        print("LLM re-ranker from_pretrained: yes-logits", want, got, "max |d|", err)
        assert err < 5e-3 * max(1.0, float(np.abs(want).max()))
    finally:
        rr.decoder.close()


@pytest.mark.gpu
def test_from_config_builds_the_hot_path_from_a_yaml_file(tmp_path):
    """The reference's YAML keys + device_ids / corpus_dtype: embeddings (a replica per listed device), re-ranker, and a retriever
    class that shards the corpus over the devices and holds it as e4m3 -- retrieve + re-rank through them end to end."""
    import yaml
    import veritasfi_amd as vf
    bt, xt = TS.bert_tokenizer(tmp_path), TS.xlmr_tokenizer()
    de = TS.write_st_dir(str(tmp_path / "emb"), bt, TS.tiny_bert(len(bt)), pooling="cls", normalize=True)
    dr = str(tmp_path / "rr")
    TS.tiny_xlmr_cross_encoder(len(xt)).save_pretrained(dr)
    xt.save_pretrained(dr)
    cfgp = tmp_path / "cfg.yaml"
    cfgp.write_text(yaml.safe_dump({"persist_directory": "unused", "embeddings_model_name": de, "rerank_model": dr, "rerank_topk": 5,
                                    "device_ids": [0, 0], "corpus_dtype": "fp8"}))
    parts = vf.from_config(str(cfgp))
    try:
        assert isinstance(parts.embeddings, vf.ReplicaSet) and len(parts.embeddings.replicas) == 2
        docs = [f"w{i} w{i + 1} revenue of w{i + 2} in the quarter" for i in range(200)]
        vecs = parts.embeddings.embed_documents(docs)
        one = parts.embeddings.replicas[0].embed_documents(docs)
        assert np.allclose(vecs, one, atol=2e-3)                               # replicas agree; order is the input's
        fr = parts.retriever_cls(vecs, parts.embeddings)
        assert fr.index.shard_devices() == [0, 0]
        I, D = fr.invoke([docs[17], docs[150]], 5)
        assert I[0, 0] == 17 and I[1, 0] == 150 and D[0, 0] > 0.97             # e4m3 rows: the stored values' cosine
        scores = parts.reranker.compute_score([[docs[17], docs[int(i)]] for i in I[0]], batch_size=8)
        assert len(scores) == 5 and np.isfinite(scores).all()
        fr.index.close()
    finally:
        parts.embeddings.close()
        parts.reranker.close()
