"""Host side of the three model loops: text -> token arrays, overlapped with the device.

The reference's loops tokenise a batch, run it, tokenise the next (``src/load_data.py:98-99,120-128`` through
``Chroma.add_texts`` -> ``embed_documents``; ``src/utils/vllmManager.py:450-452`` -> ``compute_score``), and so did this
package until round 6: the device idled while Python tokenised and the tokenizer idled while the device ran.  Two things here:

* ``BatchTokenizer`` -- a Hugging Face *fast* tokenizer is driven at its Rust backend (``encode_batch_fast`` on a private copy
  configured once with the truncation / padding the HF call would set): the ``BatchEncoding`` assembly of the Python wrapper costs
  5-8 x the tokenisation itself (100 pairs x 512 tokens, XLM-R Unigram, 8 cores: 100 ms through ``tokenizer(...)``, 13 + 4 ms
  here; same ids, masks and token types -- ``tests/test_host_tokenize.py``).  Any other tokenizer object is called the HF way.
* ``pipelined`` -- a one-deep prefetch: batch i + 1 is tokenised on a worker thread while batch i is on the device (the Rust
  tokenizer and a ctypes call both release the GIL).  The batch partition does not depend on whether the overlap is on, so the
  results are bit-equal to the serial loop's.
"""
from __future__ import annotations

import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_pool = None
_pool_mu = threading.Lock()


def _worker() -> ThreadPoolExecutor:
    """ONE shared prefetch thread pool (two workers: nested loops -- a ReplicaSet fanning batches out -- never wait on themselves)."""
    global _pool
    with _pool_mu:
        if _pool is None:
            _pool = ThreadPoolExecutor(max_workers=2, thread_name_prefix="vf-tokenize")
        return _pool


def pipelined(batches, prepare, run, overlap: bool = True, finish=None) -> list:
    """[finish(run(prepare(b))) for b in batches] with prepare(batch i + 1) -- and finish(batch i - 1), the conversion of a result into
    what the caller's interface wants (``ndarray.tolist()`` for ``embed_documents``: 2.4 ms per 100 x 768) -- running beside run(batch i)."""
    batches = list(batches)
    if not overlap or len(batches) <= 1:
        outs = [run(prepare(b)) for b in batches]
        return outs if finish is None else [finish(o) for o in outs]
    pool = _worker()
    out = []
    # The worker's Python sections (list -> array conversions) hold the interpreter lock; a thread that wants it back waits a whole
    # switch interval (5 ms by default) before it even asks.  The driving thread needs the lock for microseconds between two device
    # calls of ~10 ms: for the duration of this loop the interval is 0.2 ms (restored on every exit path).
    import sys
    keep = sys.getswitchinterval()
    sys.setswitchinterval(min(keep, 2e-4))
    # ... and the worker starts its job half a millisecond late: a tokenizer call begins (and ends) with a phase that holds the interpreter
    # lock -- Python strings in, Encoding objects out, 2-4 ms of a 100 x 512-token batch -- and a worker that starts at the same moment as
    # the driving thread takes the lock first: the device call then starts that much later, every batch (measured: a 2.6-ms hole per
    # 11-ms forward).  Half a millisecond is what the driving thread needs to get from submit() into the C call, where it needs no lock.
    def late(fn):
        def job(x):
            time.sleep(5e-4)
            return fn(x)
        return job
    prepare_l, finish_l = late(prepare), (late(finish) if finish is not None else None)
    try:
        nxt = pool.submit(prepare, batches[0])
        for i in range(len(batches)):
            ready = nxt.result()
            if i + 1 < len(batches):
                nxt = pool.submit(prepare_l, batches[i + 1])
            res = run(ready)
            out.append(res if finish is None else pool.submit(finish_l if i + 1 < len(batches) else finish, res))
        return out if finish is None else [f.result() for f in out]
    finally:
        sys.setswitchinterval(keep)


def split_for_overlap(n: int, step: int, min_piece: int = 16) -> list:
    """(lo, hi) pieces of n items in device batches of at most `step`.  A call that fits ONE device batch is cut so that tokenisation runs
    under a forward: in three pieces of about 1/4, 2/5 and the rest (whole multiples of 8) from 96 items up -- a SMALL first piece puts
    the device to work early, and the call's time tends to (tokenisation of everything) + (forward of the last piece) when the
    tokenizer is the slower side, as it is for 100 pairs of 512 tokens -- in two halves when both keep at least `min_piece` items;
    larger calls keep their `step`-sized batches (already more than one)."""
    if n <= 0:
        return []
    if n > step:
        return [(lo, min(lo + step, n)) for lo in range(0, n, step)]
    if n >= 96:
        a = max(min_piece, (n // 4) // 8 * 8)
        b = a + max(min_piece, (2 * n // 5) // 8 * 8)
        if n - b >= min_piece:
            return [(0, a), (a, b), (b, n)]
    half = (n // 2 + 7) // 8 * 8
    if half >= min_piece and n - half >= min_piece:
        return [(0, half), (half, n)]
    return [(0, n)]


def _backend(tokenizer):
    """The Rust tokenizer behind a HF fast tokenizer, or None."""
    if not getattr(tokenizer, "is_fast", False):
        return None
    return getattr(tokenizer, "backend_tokenizer", None) or getattr(tokenizer, "_tokenizer", None)


class BatchTokenizer:
    """``encode(texts)`` / ``encode(texts, second_texts)`` -> (input_ids int32 [b, t], attention_mask int32 [b, t], token_type_ids or
    None): what ``tokenizer(texts, [second,] padding=True, truncation=True, max_length=..., return_tensors="np")`` returns."""

    def __init__(self, tokenizer, max_length: int):
        self.tokenizer, self.max_length = tokenizer, int(max_length)
        self._rust = None
        self._mu = threading.Lock()
        be = _backend(tokenizer)
        if be is not None and getattr(tokenizer, "pad_token_id", None) is not None:
            try:
                from tokenizers import Tokenizer
                rust = Tokenizer.from_str(be.to_str())   # a private copy: the caller's tokenizer keeps its own truncation / padding state
                # exactly what PreTrainedTokenizerFast.set_truncation_and_padding configures for padding=True, truncation=True
                rust.enable_truncation(max_length=self.max_length, stride=0, strategy="longest_first",
                                       direction=getattr(tokenizer, "truncation_side", "right"))
                # padding to the batch's longest row is done HERE, in numpy (same result as the backend's enable_padding: pad_id,
                # pad_type_id, the tokenizer's padding side): the unpadded id lists are shorter to convert, the mask follows from the
                # lengths and a single sequence's token types are all zero -- one list -> array conversion per row instead of three,
                # and conversions are what this path costs beside the Rust call (they hold the interpreter lock)
                rust.no_padding()
                self._pad_left = getattr(tokenizer, "padding_side", "right") == "left"
                self._pad_id = int(tokenizer.pad_token_id)
                self._pad_type = int(getattr(tokenizer, "pad_token_type_id", 0) or 0)
                self._rust = rust
                self._types = "token_type_ids" in getattr(tokenizer, "model_input_names", ())
            except Exception:  # noqa: BLE001 -- an exotic tokenizer the copy cannot express: take the HF call
                self._rust = None

    @property
    def direct(self) -> bool:
        """encode() goes straight to the Rust backend (a HF fast tokenizer with a pad token)."""
        return self._rust is not None

    @property
    def rust_backed(self) -> bool:
        """The tokenizer has a Rust backend at all (encode_plain uses it, pad token or not): its work releases the interpreter lock."""
        return _backend(self.tokenizer) is not None

    def encode(self, texts, second=None):
        texts = list(texts)
        if self._rust is None:
            from .encoder import _tok_arrays
            if second is None:
                enc = self.tokenizer(texts, padding=True, truncation=True, max_length=self.max_length, return_tensors="np")
            else:
                enc = self.tokenizer(texts, list(second), padding=True, truncation=True, max_length=self.max_length, return_tensors="np")
            return _tok_arrays(enc)
        items = texts if second is None else list(zip(texts, second))
        with self._mu:   # a Rust tokenizer is Sync, but encode_batch on one object from two threads serialises on its own pool anyway
            fast = getattr(self._rust, "encode_batch_fast", None) or self._rust.encode_batch
            encs = fast(items, add_special_tokens=True)
        # row by row into preallocated arrays: ~20-us C calls with interpreter switch points between them -- one np.array() over the
        # nested lists is a single 2-ms call during which the thread that drives the device cannot take the interpreter lock back
        n = len(encs)
        rows = [e.ids for e in encs]
        lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=n)
        t = int(lens.max()) if n else 0
        ids = np.full((n, t), self._pad_id, np.int32)
        col = np.arange(t, dtype=np.int64)[None, :]
        mask = ((col >= (t - lens)[:, None]) if self._pad_left else (col < lens[:, None])).astype(np.int32)
        tt = None
        if self._types:
            tt = np.full((n, t), self._pad_type, np.int32)
            if second is None:
                tt[mask.astype(bool)] = 0            # a single sequence: every token (and its special tokens) is type 0
        for i, r in enumerate(rows):
            if self._pad_left:
                ids[i, t - len(r):] = r
            else:
                ids[i, :len(r)] = r
            if tt is not None and second is not None:
                ty = encs[i].type_ids
                if self._pad_left:
                    tt[i, t - len(ty):] = ty
                else:
                    tt[i, :len(ty)] = ty
        return ids, mask, tt

    def encode_plain(self, texts, max_length=None, add_special_tokens: bool = False):
        """Unpadded id lists, each truncated to max_length -- WITHOUT special tokens by default (the LLM re-ranker's prompt pieces):
        ``tokenizer(text, add_special_tokens=..., truncation=True, max_length=...)["input_ids"]`` for every text."""
        texts = list(texts)
        be = _backend(self.tokenizer)
        if be is None:
            kw = {} if max_length is None else {"max_length": max_length, "truncation": True}
            return [list(self.tokenizer(t, return_tensors=None, add_special_tokens=add_special_tokens, **kw)["input_ids"]) for t in texts]
        with self._mu:
            if getattr(self, "_plain", None) is None:
                from tokenizers import Tokenizer
                self._plain = Tokenizer.from_str(be.to_str())
                self._plain.no_padding()
            if max_length is None:
                self._plain.no_truncation()
            else:
                self._plain.enable_truncation(max_length=int(max_length), stride=0, strategy="longest_first",
                                              direction=getattr(self.tokenizer, "truncation_side", "right"))
            fast = getattr(self._plain, "encode_batch_fast", None) or self._plain.encode_batch
            return [list(e.ids) for e in fast(texts, add_special_tokens=bool(add_special_tokens))]
