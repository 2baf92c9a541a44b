"""Candidate gathering around the dense search: ``EnsembleRetriever`` with the reference's constructor and
``invoke(input, hyde_chunks) -> list[dict]`` (``src/utils/ensembleRetriever.py:19-48`` ctor, ``:50-232`` invoke).

What the reference does per hit with an O(N) scan over every chunk's metadata -- "all rows of this bundle"
(``:77-82,155-160,203-208``) and "all rows whose title summary is this string" (``:144``) -- is answered here from
three maps built once at construction: ``bundle_id -> rows`` (ascending row order, as the scan yields them),
``title_summary -> rows`` and ``doc_id -> row`` (the reference's ``docid2idx``, ``:45``).  The dense searches go
through this package's ``FaissRetriever`` (GPU).  Output schema, ordering, the ``seen_ids`` bookkeeping, the
neighbour expansion (score > 0.72, neighbours > 0.66 in the 2048-deep score map, at most 4 rows, ``:85-107``) and
the running ``bundle_id`` counter are the reference's.

BM25 itself is outside this path (CPU, sparse): the constructor does what the reference's does (``:37``) -- it builds the
host application's own ``BM25Retriever(bm25_dir)`` -- unless an object with ``invoke(query, k) -> (ids, scores)``
(``src/utils/bm25Retriever.py:50-87``) is handed in as ``bm25_retriever``.  When the BM25 branch is on (``bm25_k`` resolves
to > 0) and neither works, construction FAILS; the branch is never dropped silently.  The stores are anything with
Chroma's ``get(include=[...])`` / ``get(ids=[...], include=[...])``.
"""
from __future__ import annotations

import importlib
import threading
from collections import OrderedDict
from typing import Dict, List

import numpy as np

from . import stages
from .faiss_retriever import FaissRetriever
from .similarity import _embed_all, compute_similarity, compute_similarity_mtx

EXPAND_SCORE, NEIGHBOUR_SCORE, MAX_EXPANDED, SEARCH_DEPTH = 0.72, 0.66, 4, 2048
# where the host application's BM25Retriever lives, as seen from its entry points (src/ on sys.path, or the repo root)
BM25_MODULES = ("utils.bm25Retriever", "src.utils.bm25Retriever", "bm25Retriever")
TEXT_ROWS_KEPT = 65536   # chunk texts whose corpus row is remembered (a request emits tens to ~155)


class _TextRows:
    """chunk text -> corpus row, for the texts this retriever has emitted (bounded, least recently emitted first out).
    Request threads share a retriever without a lock upstream (ragManager.py:17-30), so this has its own."""

    def __init__(self, cap: int = TEXT_ROWS_KEPT):
        self._cap, self._map, self._mu = cap, OrderedDict(), threading.Lock()

    def remember(self, texts, rows) -> None:
        with self._mu:
            for text, row in zip(texts, rows):
                if row is None or row < 0:
                    continue
                self._map[text] = row
                self._map.move_to_end(text)
            while len(self._map) > self._cap:
                self._map.popitem(last=False)

    def rows_of(self, texts) -> List[int]:
        with self._mu:
            return [self._map.get(t, -1) for t in texts]


def _host_bm25(bm25_dir):
    """``BM25Retriever(bm25_dir)`` of the application this class is dropped into (ensembleRetriever.py:13,37)."""
    tried = []
    for name in BM25_MODULES:
        try:
            mod = importlib.import_module(name)
        except ImportError as e:
            tried.append(f"{name}: {e}")
            continue
        return mod.BM25Retriever(bm25_dir)
    raise ImportError("EnsembleRetriever: the BM25 branch is on (bm25_k > 0) but no BM25Retriever is importable ("
                      + "; ".join(tried) + ") -- pass bm25_retriever=<object with invoke(query, k)>, or bm25_k=0 to "
                      "switch the branch off explicitly")


class EnsembleRetriever:
    def __init__(self, bm25_dir, chroma, ts_chroma, k: int, embeddings, faiss_k: int = None, bm25_k: int = None,
                 faiss_ts_k: int = None, enable_expand: bool = False, bm25_retriever=None, retriever_cls=FaissRetriever,
                 prefetch_documents: bool = False, similarity_from_rows: bool = None):
        """Positional arguments as upstream (``ragManager.py:112`` passes bm25_dir, chroma, ts_chroma, k, embeddings).
        ``bm25_dir`` goes to the host application's ``BM25Retriever`` exactly as upstream (``:37``) unless
        ``bm25_retriever`` is given.
        ``prefetch_documents=True`` loads every chunk's text once and serves bundles from memory in the requested
        row order instead of one ``chroma.get(ids=...)`` per bundle (Chroma returns rows in ITS order, so leave this
        off when byte-identical ordering inside a bundle matters).
        ``similarity_from_rows``: ``compute_similarity_mtx(texts)`` -- the reference's call, texts only (vllmManager.py:462) --
        serves every text this retriever has emitted from ITS ROW of the HBM-resident corpus (``vf_cosine_matrix_rows_mixed``) and
        embeds only texts it does not know, instead of embedding all n again (ensembleRetriever.py:275).  None (default) = on when
        that is the same matrix: the rows are held as given (not rounded to a narrower ``corpus_dtype``) and the embedder has no
        query instruction (upstream stores ``embed_documents(text)`` and compares ``embed_query(text)``: the same vector unless
        the embedder prefixes queries).  False = always re-embed; True = rows even where auto would decline."""
        self.embeddings = embeddings
        self.faiss_k = faiss_k if faiss_k is not None else k
        self.bm25_k = bm25_k if bm25_k is not None else k
        self.faiss_ts_k = faiss_ts_k if faiss_ts_k is not None else k
        self.enable_expand = enable_expand
        self.chroma = chroma
        self.bm25_dir = bm25_dir
        self.bm25_retriever = bm25_retriever
        if self.bm25_retriever is None and self.bm25_k > 0:
            self.bm25_retriever = _host_bm25(bm25_dir)

        include = ["metadatas", "embeddings"] + (["documents"] if prefetch_documents else [])
        docs = chroma.get(include=include)
        self.faiss_retriever = retriever_cls(docs["embeddings"], embeddings)
        ts_docs = ts_chroma.get(include=["documents", "embeddings"])
        self.title_summary_faiss_retriever = retriever_cls(ts_docs["embeddings"], embeddings)

        self.chunk_metadata = docs["metadatas"]
        self.num_chunk = len(self.chunk_metadata)
        self.title_summaries = ts_docs["documents"]
        self._documents = docs["documents"] if prefetch_documents else None
        # the three maps that replace the per-hit scans
        self.docid2idx: Dict[str, int] = {}
        self._bundle_rows: Dict[object, List[int]] = {}
        self._title_rows: Dict[str, List[int]] = {}
        for row, md in enumerate(self.chunk_metadata):
            self.docid2idx[md["doc_id"]] = row
            b = md.get("bundle_id", None)
            if b is not None:
                self._bundle_rows.setdefault(b, []).append(row)
            self._title_rows.setdefault(md.get("title_summary", ""), []).append(row)
        self._text_rows = _TextRows()
        self.similarity_from_rows = self._rows_serve_similarity() if similarity_from_rows is None else bool(similarity_from_rows)

    def _rows_serve_similarity(self) -> bool:
        ix = getattr(self.faiss_retriever, "index", None)
        if ix is None or not hasattr(ix, "cosine_matrix_rows"):
            return False
        if not getattr(self.faiss_retriever, "rows_as_given", False):
            return False
        for name in ("query_instruction", "query_instruction_for_retrieval", "embed_instruction"):
            if getattr(self.embeddings, name, None):
                return False
        return True

    # -- pieces -------------------------------------------------------------------------------------
    def _bundle_of(self, row: int, seen: set) -> List[int]:
        """Rows emitted for a hit: the whole bundle when the chunk has one (all marked seen), else the row."""
        b = self.chunk_metadata[row].get("bundle_id", None)
        if b is None:
            return [row]
        rows = list(self._bundle_rows[b])
        seen.update(rows)
        return rows

    def _fetch(self, rows: List[int]):
        if self._documents is not None:
            return [self._documents[r] for r in rows], [self.chunk_metadata[r] for r in rows]
        got = self.chroma.get(ids=[self.chunk_metadata[r]["doc_id"] for r in rows], include=["documents", "metadatas"])
        return got["documents"], got["metadatas"]

    def _emit(self, out: list, name: str, score, rows: List[int], bundle_cnt: int) -> None:
        documents, metadatas = self._fetch(rows)
        if self.similarity_from_rows:   # the store answers in ITS order: a text's row comes from its own metadata
            self._text_rows.remember(documents, [self.docid2idx.get(md.get("doc_id"), -1) for md in metadatas])
        for text, md in zip(documents, metadatas):
            out.append({"retriever": name, "score": float(score), "page_content": text, "metadata": md,
                        "bundle_id": bundle_cnt})

    def _expand(self, rows: List[int], hit_md: dict, score_map: dict, seen: set) -> None:
        prev_doc, next_doc = hit_md["prev_chunk_id"], hit_md["next_chunk_id"]
        while len(rows) < MAX_EXPANDED:
            grew = False
            p = self.docid2idx.get(prev_doc, -1) if prev_doc != "" else -1
            if p != -1 and score_map.get(p, 0) > NEIGHBOUR_SCORE and p not in seen:
                grew = True
                seen.add(p)
                rows.insert(0, p)
                prev_doc = self.chunk_metadata[p]["prev_chunk_id"]
            n = self.docid2idx.get(next_doc, -1) if next_doc != "" else -1
            if n != -1 and score_map.get(n, 0) > NEIGHBOUR_SCORE and n not in seen:
                grew = True
                seen.add(n)
                rows.append(n)
                next_doc = self.chunk_metadata[n]["next_chunk_id"]
            if not grew:
                break

    # -- the reference's entry point ------------------------------------------------------------------
    def invoke(self, input: str, hyde_chunks: List[str]) -> List[Dict]:
        """Stage brackets as upstream (``"retrieve"`` around the call, one per branch, the ``retrieved_chunks`` metric): no-ops
        unless ``veritasfi_amd.set_profiler`` was given the host's profiler (stages.py)."""
        with stages.stage("retrieve"):
            out = self._invoke(input, hyde_chunks)
        stages.metric("retrieved_chunks", len(out))
        return out

    def _invoke(self, input: str, hyde_chunks: List[str]) -> List[Dict]:
        seen: set = set()
        out: list = []
        bundle_cnt = 0
        if self.faiss_k > 0:
            with stages.stage("retrieve_faiss"):
                bundle_cnt = self._faiss_branch(input, hyde_chunks, seen, out, bundle_cnt)
        if self.faiss_ts_k > 0:
            with stages.stage("retrieve_faiss_ts"):
                bundle_cnt = self._title_branch(input, seen, out, bundle_cnt)
        if self.bm25_k > 0:
            with stages.stage("retrieve_bm25"):
                bundle_cnt = self._bm25_branch(input, seen, out, bundle_cnt)
        return out

    def _faiss_branch(self, input, hyde_chunks, seen, out, bundle_cnt):
        inputs = [input] + list(hyde_chunks)
        # upstream always searches 2048 deep (:64-66) and reads the tail only as the neighbour expansion's score map (:85-107): without
        # enable_expand the first faiss_k entries are all that is looked at, and an exact search's top-k is a prefix of its top-2048
        depth = SEARCH_DEPTH if self.enable_expand else min(SEARCH_DEPTH, max(1, int(self.faiss_k)))
        ids_list, scores_list = self.faiss_retriever.invoke(inputs, depth)
        for ids, scores in zip(ids_list, scores_list):
            ids = [int(i) for i in ids]
            score_map = dict(zip(ids, scores))
            for row, score in zip(ids[:self.faiss_k], scores[:self.faiss_k]):
                if row < 0 or row in seen:   # -1 pads a corpus smaller than the search depth
                    continue
                seen.add(row)
                md = self.chunk_metadata[row]
                rows = self._bundle_of(row, seen)
                if score > EXPAND_SCORE and self.enable_expand:
                    self._expand(rows, md, score_map, seen)
                self._emit(out, "FAISS", score, rows, bundle_cnt)
                bundle_cnt += 1
        return bundle_cnt

    def _title_branch(self, input, seen, out, bundle_cnt):
        t_ids, t_scores = self.title_summary_faiss_retriever.invoke([input], self.faiss_ts_k)
        for t, score in zip(t_ids[0], t_scores[0]):
            if t < 0:
                continue
            for row in self._title_rows.get(self.title_summaries[int(t)], ()):
                if row in seen:
                    continue
                seen.add(row)
                self._emit(out, "Title Summary", score, self._bundle_of(row, seen), bundle_cnt)
                bundle_cnt += 1
        return bundle_cnt

    def _bm25_branch(self, input, seen, out, bundle_cnt):
        b_ids, b_scores = self.bm25_retriever.invoke(input, self.num_chunk)
        for row, score in zip(b_ids[:self.bm25_k], b_scores[:self.bm25_k]):
            if row in seen:
                continue
            seen.add(row)
            self._emit(out, "BM25", score, self._bundle_of(row, seen), bundle_cnt)
            bundle_cnt += 1
        return bundle_cnt

    # ensembleRetriever.py:235-281
    def compute_similarity(self, chunks, selected_indices, candidate_index):
        return compute_similarity(self.embeddings, chunks, selected_indices, candidate_index)

    def compute_similarity_mtx(self, chunks):
        """The reference's signature and return (``[n, n]`` tensor indexed ``similar_mtx[idx, selected] > 0.9``,
        vllmManager.py:462,476).  Texts this retriever emitted are read from their corpus rows in HBM; the others are embedded in
        ONE batched call; with ``similarity_from_rows`` off (or nothing known) every text is embedded, as upstream."""
        chunks = list(chunks)
        rows = self._text_rows.rows_of(chunks) if self.similarity_from_rows else []
        if not chunks or not any(r >= 0 for r in rows):
            return compute_similarity_mtx(self.embeddings, chunks)
        unknown = [t for t, r in zip(chunks, rows) if r < 0]
        extra = _embed_all(self.embeddings, unknown) if unknown else None
        ix = self.faiss_retriever.index
        ids = np.asarray([r + ix.id_offset if r >= 0 else -1 for r in rows], dtype=np.int64)
        import torch
        return torch.from_numpy(ix.cosine_matrix_rows(ids, extra))
