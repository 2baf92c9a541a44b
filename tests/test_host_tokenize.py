"""Host tokenisation pipeline (veritasfi_amd/host_tokenize.py): the Rust-backend route returns exactly what the Hugging Face call
returns for the three tokenizer families of the path, and the prefetch loop keeps order and results.  CPU only."""
import threading
import time

import numpy as np
import pytest

import tokenizers_synth as TS
from veritasfi_amd.host_tokenize import BatchTokenizer, pipelined, split_for_overlap


def _texts(seed, n, lo=3, hi=90):
    rng = np.random.default_rng(seed)
    return [" ".join(TS.WORDS[i] for i in rng.integers(0, len(TS.WORDS), int(rng.integers(lo, hi)))) for _ in range(n)]


@pytest.mark.parametrize("family", ["bert", "xlmr", "gemma"])
@pytest.mark.parametrize("max_length", [16, 64])
def test_rust_backend_route_equals_the_hf_call(tmp_path, family, max_length):
    tok = {"bert": lambda: TS.bert_tokenizer(tmp_path), "xlmr": TS.xlmr_tokenizer, "gemma": TS.gemma_tokenizer}[family]()
    bt = BatchTokenizer(tok, max_length)
    assert bt.direct, "a fast tokenizer must take the Rust route"
    a, b = _texts(1, 23), _texts(2, 23)
    before = (tok.backend_tokenizer.padding, tok.backend_tokenizer.truncation)
    bt.encode(a, b)
    assert (tok.backend_tokenizer.padding, tok.backend_tokenizer.truncation) == before, \
        "the caller's tokenizer keeps its own state: the route configures a private copy"
    for first, second in ((a, None), (a, b), (a[:1], None), (a[:1], b[:1])):
        want = tok(first, padding=True, truncation=True, max_length=max_length, return_tensors="np") if second is None else \
            tok(first, second, padding=True, truncation=True, max_length=max_length, return_tensors="np")
        ids, mask, tt = bt.encode(first, second)
        assert ids.dtype == np.int32 and np.array_equal(ids, want["input_ids"]) and np.array_equal(mask, want["attention_mask"])
        if "token_type_ids" in want:
            assert np.array_equal(tt, want["token_type_ids"])
        else:
            assert tt is None
        assert ids.shape[1] <= max_length
    plain = bt.encode_plain(a, 12)
    assert plain == [list(tok(t, add_special_tokens=False, truncation=True, max_length=12)["input_ids"]) for t in a]
    special = bt.encode_plain(a, 12, add_special_tokens=True)
    assert special == [list(tok(t, truncation=True, max_length=12)["input_ids"]) for t in a]
    assert bt.encode_plain(a[:3]) == [list(tok(t, add_special_tokens=False)["input_ids"]) for t in a[:3]]


def test_objects_that_are_not_fast_tokenizers_are_called_the_hf_way():
    calls = []

    class Plain:
        def __call__(self, a, b=None, padding=None, truncation=None, max_length=None, return_tensors=None):
            calls.append((tuple(a), None if b is None else tuple(b), padding, truncation, max_length, return_tensors))
            n = len(a)
            return {"input_ids": np.ones((n, 4), np.int64), "attention_mask": np.ones((n, 4), np.int64)}

    bt = BatchTokenizer(Plain(), 32)
    assert not bt.direct
    ids, mask, tt = bt.encode(["x", "y"], ["u", "v"])
    assert ids.shape == (2, 4) and tt is None and calls == [(("x", "y"), ("u", "v"), True, True, 32, "np")]


def test_split_for_overlap():
    assert split_for_overlap(0, 128) == []
    assert split_for_overlap(100, 128) == [(0, 24), (24, 64), (64, 100)]   # the reference's 100 pairs: a small first piece, then two larger ones
    assert split_for_overlap(64, 128) == [(0, 32), (32, 64)]               # two halves below 96
    assert split_for_overlap(13, 128) == [(0, 13)]                          # a data-parallel share stays whole
    assert split_for_overlap(300, 128) == [(0, 128), (128, 256), (256, 300)]
    assert split_for_overlap(100, 32) == [(0, 32), (32, 64), (64, 96), (96, 100)]
    for n in range(1, 200):
        for step in (8, 32, 128):
            p = split_for_overlap(n, step)
            assert p[0][0] == 0 and p[-1][1] == n and all(a[1] == b[0] for a, b in zip(p, p[1:])) and all(hi - lo <= step for lo, hi in p)


def test_pipelined_keeps_order_and_overlaps_prepare_with_run():
    log, lock = [], threading.Lock()

    def prepare(b):
        with lock:
            log.append(("p+", b))
        time.sleep(0.03)
        with lock:
            log.append(("p-", b))
        return b * 10

    def run(x):
        with lock:
            log.append(("r+", x))
        time.sleep(0.03)
        with lock:
            log.append(("r-", x))
        return x + 1

    assert pipelined([1, 2, 3, 4], prepare, run) == [11, 21, 31, 41]
    # the next batch's prepare started before the current batch's run ended
    assert log.index(("p+", 2)) < log.index(("r-", 10)) and log.index(("p+", 3)) < log.index(("r-", 20))
    t0 = time.perf_counter()
    pipelined(list(range(6)), prepare, run)
    overlapped = time.perf_counter() - t0
    t0 = time.perf_counter()
    assert pipelined(list(range(6)), prepare, run, overlap=False) == [b * 10 + 1 for b in range(6)]
    serial = time.perf_counter() - t0
    assert overlapped < 0.8 * serial
    assert pipelined([], prepare, run) == [] and pipelined([7], prepare, run) == [71]
    with pytest.raises(ZeroDivisionError):      # a failure in the worker surfaces in the caller
        pipelined([1, 0, 2], lambda b: 1 // b, lambda x: x)
