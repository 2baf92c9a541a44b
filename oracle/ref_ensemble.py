"""Literal restatement of ``EnsembleRetriever.invoke`` (``src/utils/ensembleRetriever.py:50-232``) -- TEST INFRASTRUCTURE ONLY.

Keeps the reference's per-hit linear scans over ``chunk_metadata`` (bundle members ``:81,159,207``; title-summary
members ``:144``), its ``seen_ids`` updates, the neighbour expansion loop (``:85-107``) and the order in which dicts
are appended, so that the pre-indexed product class can be compared with it on the same retriever outputs.
``dense`` / ``ts_dense`` / ``bm25`` are callables standing for the three retrievers' ``invoke``.
"""


def gather(input, hyde_chunks, *, chunk_metadata, title_summaries, store_get, dense, ts_dense, bm25,
           faiss_k, faiss_ts_k, bm25_k, enable_expand):
    docid2idx = {m["doc_id"]: i for i, m in enumerate(chunk_metadata)}                 # :45
    seen_ids, chunk_list, bundle_cnt = set(), [], 0

    def bundle_scan(idx):
        md = chunk_metadata[idx]
        if md.get("bundle_id", None) != None:  # noqa: E711  (the reference's comparison)
            members = [i for i, m in enumerate(chunk_metadata) if m.get("bundle_id", None) == md["bundle_id"]]
            seen_ids.update(members)
            return members
        return [idx]

    def append(name, score, ids):
        got = store_get([chunk_metadata[i]["doc_id"] for i in ids])
        for j in range(len(got["documents"])):
            chunk_list.append({"retriever": name, "score": float(score), "page_content": got["documents"][j],
                               "metadata": got["metadatas"][j], "bundle_id": bundle_cnt})

    if faiss_k > 0:
        inputs = [input] + hyde_chunks
        ids_list, scores_list = dense(inputs, 2048)                                     # :66
        for faiss_ids, faiss_scores in zip(ids_list, scores_list):
            effective = {i: s for i, s in zip(faiss_ids, faiss_scores)}                 # :68
            for idx, score in zip(faiss_ids[:faiss_k], faiss_scores[:faiss_k]):
                if idx in seen_ids:
                    continue
                seen_ids.add(idx)
                md = chunk_metadata[idx]
                ids = bundle_scan(idx)
                if (score > 0.72) and enable_expand:                                    # :85
                    prev_doc_id, next_doc_id = md["prev_chunk_id"], md["next_chunk_id"]
                    while len(ids) < 4:
                        flag = False
                        if prev_doc_id != "" and docid2idx.get(prev_doc_id, -1) != -1:
                            prev_id = docid2idx[prev_doc_id]
                            if effective.get(prev_id, 0) > 0.66 and prev_id not in seen_ids:
                                flag = True
                                seen_ids.add(prev_id)
                                ids.insert(0, prev_id)
                                prev_doc_id = chunk_metadata[prev_id]["prev_chunk_id"]
                        if next_doc_id != "" and docid2idx.get(next_doc_id, -1) != -1:
                            next_id = docid2idx[next_doc_id]
                            if effective.get(next_id, 0) > 0.66 and next_id not in seen_ids:
                                flag = True
                                seen_ids.add(next_id)
                                ids.append(next_id)
                                next_doc_id = chunk_metadata[next_id]["next_chunk_id"]
                        if not flag:
                            break
                append("FAISS", score, ids)
                bundle_cnt += 1
    if faiss_ts_k > 0:
        t_ids, t_scores = ts_dense([input], faiss_ts_k)                                 # :139
        for title_idx, score in zip(t_ids[0], t_scores[0]):
            title = title_summaries[title_idx]
            for idx in [i for i, m in enumerate(chunk_metadata) if m.get("title_summary", "") == title]:   # :144
                if idx in seen_ids:
                    continue
                seen_ids.add(idx)
                append("Title Summary", score, bundle_scan(idx))
                bundle_cnt += 1
    if bm25_k > 0:
        b_ids, b_scores = bm25(input, len(chunk_metadata))                              # :190
        for idx, score in zip(b_ids[:bm25_k], b_scores[:bm25_k]):
            if idx in seen_ids:
                continue
            seen_ids.add(idx)
            append("BM25", score, bundle_scan(idx))
            bundle_cnt += 1
    return chunk_list
