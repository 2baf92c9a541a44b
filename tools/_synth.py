"""Synthetic stand-ins shared by the harnesses: there are no tokenizer files or checkpoints offline, so texts are
tokenised by a deterministic whitespace hash and models are random-init encoders of a named shape."""
import itertools

import numpy as np

WORDS = ["revenue", "segment", "filing", "quarter", "margin", "vehicle", "delivery", "guidance", "capex", "cash",
         "battery", "forecast", "dividend", "liability", "auditor", "subsidiary", "tariff", "inventory", "lease", "equity"]


class HashTokenizer:
    """HF call signature (texts[, pairs], padding, truncation, max_length, return_tensors) -> input_ids / attention_mask."""

    def __init__(self, vocab: int, bos: int = 0, sep: int = 2, pad: int = 1):
        self.vocab, self.bos, self.sep, self.pad = vocab, bos, sep, pad
        self._cache = {}

    def _tok(self, w):
        t = self._cache.get(w)
        if t is None:
            t = self._cache[w] = self._hash(w)
        return t

    def _hash(self, w):
        h = 2166136261
        for c in w.encode():
            h = ((h ^ c) * 16777619) & 0xFFFFFFFF
        return 5 + h % (self.vocab - 5)

    def __call__(self, a, b=None, padding=True, truncation=True, max_length=512, return_tensors="np", **_):
        a = [a] if isinstance(a, str) else list(a)
        b = [None] * len(a) if b is None else ([b] if isinstance(b, str) else list(b))
        rows = []
        cache, slow = self._cache, self._tok
        def toks(text):           # one dict lookup per word; the Python-level hash runs once per distinct word
            words = text.split()
            try:
                return list(map(cache.__getitem__, words))
            except KeyError:
                return [cache[w] if w in cache else slow(w) for w in words]
        for x, y in zip(a, b):
            t = [self.bos] + toks(x) + [self.sep]
            if y is not None:
                t += [self.sep] + toks(y)[:max_length] + [self.sep]
            rows.append(t[:max_length])
        width = max(len(r) for r in rows)
        ids = np.full((len(rows), width), self.pad, np.int64)
        mask = np.zeros((len(rows), width), np.int64)
        lens = np.fromiter(map(len, rows), np.int64, len(rows))
        flat = np.fromiter(itertools.chain.from_iterable(rows), np.int64, int(lens.sum()))
        keep = np.arange(width)[None, :] < lens[:, None]
        ids[keep] = flat
        mask[keep] = 1
        if return_tensors == "pt":
            import torch
            return {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask)}
        return {"input_ids": ids, "attention_mask": mask}


def sentence(rng, n_words):
    return " ".join(WORDS[i] + str(int(j)) for i, j in zip(rng.integers(0, len(WORDS), n_words), rng.integers(0, 50, n_words)))


class LLMHashTokenizer:
    """What ``build_llm_reranker_inputs`` / ``HipLLMReranker`` need of a decoder re-ranker's tokenizer (the reference's
    get_inputs calls, stress_test.py:97-146): ``tok(text, add_special_tokens=False, max_length=, truncation=)`` ->
    ``{"input_ids": [...]}``, ``bos_token_id``, ``pad_token_id``; left padding is the re-ranker's own."""
    bos_token_id, pad_token_id, padding_side = 2, 0, "left"

    def __init__(self, vocab: int):
        self._h = HashTokenizer(vocab)

    def __call__(self, text, return_tensors=None, add_special_tokens=False, max_length=None, truncation=False, **_):
        ids = [4] if text == "\n" else [self._h._tok(w) for w in text.replace("\n", " \n ").split(" ") if w != ""]
        if truncation and max_length is not None:
            ids = ids[:max_length]
        return {"input_ids": ids}
