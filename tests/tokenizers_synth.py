"""Real HF fast tokenizers over synthetic vocabularies (no vocabulary files exist offline; the `tokenizers` package builds them):
the three families the reference's path uses -- WordPiece with token types (BERT / bge-base, -large), Unigram with the XLM-R special
ids and pair template (bge-m3, bge-reranker-base / -large) and a left-padding decoder tokenizer with a bos token (gemma /
bge-reranker-v2-gemma, /root/reference/config/example.yaml:9) -- plus tiny sentence-transformers-layout and transformers-layout
model directories written to disk for the from_pretrained / from_config tests."""
import json
import os

WORDS = [f"w{i}" for i in range(300)] + ("the of and to in a is that for it as was with be by on not he this are or his from at which but have an "
                                          "had they you were their one all we can her has there been if more when will would who so no revenue "
                                          "margin quarter fiscal deliveries guidance cash flow table figure").split()


def bert_tokenizer(tmp):
    from transformers import BertTokenizerFast
    vocab = ["[PAD]", "[unused0]", "[CLS]", "[SEP]", "[UNK]", "[MASK]"] + WORDS + ["##s", "##ing", "##ed", "##ly"]
    path = os.path.join(str(tmp), "bert_vocab.txt")
    with open(path, "w") as f:
        f.write("\n".join(vocab))
    return BertTokenizerFast(path, do_lower_case=True)


def xlmr_tokenizer():
    from tokenizers import Tokenizer, decoders, models, normalizers, pre_tokenizers, processors
    from transformers import XLMRobertaTokenizerFast
    pieces = [("<s>", 0.0), ("<pad>", 0.0), ("</s>", 0.0), ("<unk>", 0.0)] + [("▁" + w, -5.0 - 0.01 * i) for i, w in enumerate(WORDS)] + \
             [(c, -12.0) for c in "abcdefghijklmnopqrstuvwxyz0123456789"] + [("▁", -8.0)]
    tk = Tokenizer(models.Unigram(pieces, unk_id=3))
    tk.normalizer = normalizers.Sequence([normalizers.Replace("  ", " ")])
    tk.pre_tokenizer = pre_tokenizers.Metaspace()
    tk.decoder = decoders.Metaspace()
    tk.post_processor = processors.TemplateProcessing(single="<s> $A </s>", pair="<s> $A </s> </s> $B </s>", special_tokens=[("<s>", 0), ("</s>", 2)])
    return XLMRobertaTokenizerFast(tokenizer_object=tk, bos_token="<s>", eos_token="</s>", unk_token="<unk>", pad_token="<pad>", cls_token="<s>",
                                   sep_token="</s>", mask_token="<mask>")


def gemma_tokenizer():
    """word-level, <bos> in front, LEFT padding; "Yes", "\\n", "A:", "B:" are tokens, as the LLM re-ranker's prompt needs them"""
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    wl = {"<pad>": 0, "<eos>": 1, "<bos>": 2, "<unk>": 3, "Yes": 4, "No": 5, "\n": 6, "A:": 7, "B:": 8}
    for w in WORDS + ("Given query passage determine whether the contains an answer by providing prediction of either 'Yes' or 'No'. A B, "
                      "a and to").split():
        wl.setdefault(w, len(wl))
    tk = Tokenizer(models.WordLevel(wl, unk_token="<unk>"))
    tk.pre_tokenizer = pre_tokenizers.Split(" ", "removed")
    tk.post_processor = processors.TemplateProcessing(single="<bos> $A", pair="<bos> $A $B", special_tokens=[("<bos>", 2)])
    return PreTrainedTokenizerFast(tokenizer_object=tk, bos_token="<bos>", eos_token="<eos>", unk_token="<unk>", pad_token="<pad>", padding_side="left")


def _half(model):
    return model.eval().half().float()   # both sides of a parity test see the same fp16-representable weights


def write_st_dir(path, tok, model, pooling="cls", normalize=True, max_seq_length=64, modules=True):
    """A sentence-transformers model directory: transformer + tokenizer at the root, 1_Pooling/config.json, 2_Normalize, modules.json
    (``modules=False``: a plain transformers checkpoint, which sentence-transformers mean-pools)."""
    os.makedirs(path, exist_ok=True)
    model.save_pretrained(path)
    tok.save_pretrained(path)
    if not modules:
        return path
    mods = [{"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
            {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"}]
    os.makedirs(os.path.join(path, "1_Pooling"), exist_ok=True)
    json.dump({"word_embedding_dimension": model.config.hidden_size, "pooling_mode_cls_token": pooling == "cls",
               "pooling_mode_mean_tokens": pooling == "mean", "pooling_mode_max_tokens": False, "pooling_mode_mean_sqrt_len_tokens": False,
               "pooling_mode_weightedmean_tokens": False, "pooling_mode_lasttoken": pooling == "lasttoken", "include_prompt": True},
              open(os.path.join(path, "1_Pooling", "config.json"), "w"))
    if normalize:
        os.makedirs(os.path.join(path, "2_Normalize"), exist_ok=True)
        mods.append({"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"})
    json.dump(mods, open(os.path.join(path, "modules.json"), "w"))
    json.dump({"max_seq_length": max_seq_length, "do_lower_case": False}, open(os.path.join(path, "sentence_bert_config.json"), "w"))
    return path


def tiny_bert(vocab, seed=0, type_vocab=2):
    import torch
    from transformers import BertConfig, BertModel
    torch.manual_seed(seed)
    return _half(BertModel(BertConfig(vocab_size=vocab, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
                                      max_position_embeddings=128, type_vocab_size=type_vocab), add_pooling_layer=False))


def tiny_xlmr_cross_encoder(vocab, seed=1):
    import torch
    from transformers import XLMRobertaConfig, XLMRobertaForSequenceClassification
    torch.manual_seed(seed)
    m = XLMRobertaForSequenceClassification(XLMRobertaConfig(vocab_size=vocab, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                                                             intermediate_size=512, max_position_embeddings=130, type_vocab_size=1, num_labels=1,
                                                             pad_token_id=1, bos_token_id=0, eos_token_id=2))
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
        m.classifier.out_proj.weight.mul_(6.0)
    return _half(m)


def tiny_gemma_lm(vocab, seed=2):
    import torch
    from transformers import GemmaConfig, GemmaForCausalLM
    torch.manual_seed(seed)
    return _half(GemmaForCausalLM(GemmaConfig(vocab_size=vocab, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                                              head_dim=64, intermediate_size=256, max_position_embeddings=512, pad_token_id=0, bos_token_id=2,
                                              eos_token_id=1, hidden_activation="gelu_pytorch_tanh")))


def tiny_qwen3(vocab, seed=3):
    import torch
    from transformers import Qwen3Config, Qwen3Model
    torch.manual_seed(seed)
    return _half(Qwen3Model(Qwen3Config(vocab_size=vocab, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                                        head_dim=64, intermediate_size=256, max_position_embeddings=512, pad_token_id=0, bos_token_id=2,
                                        eos_token_id=1, tie_word_embeddings=True)))
