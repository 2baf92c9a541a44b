"""GPU parity tests: the HIP path through the C ABI vs the CPU oracle, bit for bit.

Run on the MI355X box:  python -m pytest tests -m gpu -q
Bar (BASELINE.json north_star): ids and rank order identical to the CPU path, scores within 1e-3 --
here scores are required to be BIT-identical to the canonical oracle, which is stronger.
"""
import os
import threading

import numpy as np
import pytest

import golden_inputs as GI
from conftest import assert_ranked, assert_topk_equiv, load_golden

pytestmark = pytest.mark.gpu

FLT_MAX = np.finfo(np.float32).max
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def vf():
    import veritasfi_amd as m
    from veritasfi_amd import _ffi
    _ffi.lib()  # raises if the HIP library is missing: no fallback
    n = _ffi.c_i32(0)
    _ffi.check(_ffi.lib().vf_device_count(n), "vf_device_count")
    assert n.value >= 1, "no GPU visible"
    return m


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _assert_exact(oracle, corpus, queries, k, ids, scores, id_offset=0):
    oi, os_ = oracle.search(corpus, queries, k, id_offset=id_offset)
    bad = np.nonzero((oi != ids).any(axis=1))[0]
    assert bad.size == 0, f"ids differ for queries {bad[:8].tolist()} (first: got {ids[bad[0]][:8]}, want {oi[bad[0]][:8]})"
    assert np.array_equal(_bits(os_), _bits(scores)), float(np.max(np.abs(os_ - scores)))
    for q in range(ids.shape[0]):
        assert_ranked(ids[q], scores[q])


def _data(seed, n, d, nq, dtype):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((n, d)).astype(np.float32)
    if dtype == np.float16:
        c = c.astype(np.float16)
    q = np.random.default_rng(seed + 1).standard_normal((nq, d)).astype(np.float32)
    return c, q


# ---- small-N exact dense path ---------------------------------------------------------------------
@pytest.mark.parametrize("n,d,nq,k,dtype", [
    (1000, 768, 4, 10, np.float32),
    (7, 32, 2, 10, np.float32),         # k > n: -1 / -FLT_MAX padding
    (150, 768, 1, 3, np.float16),
    (513, 100, 9, 40, np.float32),      # d not a multiple of 16
    (64, 1, 3, 5, np.float32),          # d = 1
    (16384, 256, 70, 100, np.float16),  # largest small-path corpus, two query batches
    (2048, 1024, 3, 2048, np.float32),  # the reference's k = 2048 (ensembleRetriever.py:66), all rows ranked
])
def test_small_path_bit_exact(vf, oracle, n, d, nq, k, dtype):
    c, q = _data(11, n, d, nq, dtype)
    with vf.DenseIndex(c) as ix:
        ids, sc = ix.search(q, k)
        assert ix.stats()["path"] == 0
    _assert_exact(oracle, c, q, k, ids, sc)
    if k > n:
        assert np.all(ids[:, n:] == -1) and np.all(sc[:, n:] == -FLT_MAX)


@pytest.mark.parametrize("n,d,nq,k", [
    (10000, 1024, 4, 2048),   # the reference's per-request shape (ensembleRetriever.py:64-66): radix select + 2048-key sort
    (10000, 64, 5, 1),
    (16384, 128, 3, 2048),    # largest corpus the select kernel holds in LDS
    (3000, 96, 2, 1500),      # k = n / 2: last shape that takes the select
    (3000, 96, 2, 1501),      # ... and the first that takes the full sort
    (9000, 32, 6, 257),       # k just past a power of two
])
def test_small_path_topk_select_with_ties(vf, oracle, n, d, nq, k):
    """k_topk_rows (radix select of the k-th key, compaction, sort of the selected keys) against the oracle on data full
    of exact ties: a third of the rows are duplicates of other rows (equal scores: the lower id must win, at the k-th
    boundary too), some rows are zero, one query is a corpus row (a score of exactly 1)."""
    c, q = _data(23, n, d, nq, np.float32)
    rng = np.random.default_rng(5)
    dup = rng.integers(0, n, size=n // 3)
    src = rng.integers(0, n, size=n // 3)
    c[dup] = c[src]
    c[rng.integers(0, n, size=7)] = 0
    q[0] = c[n // 2]
    with vf.DenseIndex(c) as ix:
        ids, sc = ix.search(q, k)
        assert ix.stats()["path"] == 0
    _assert_exact(oracle, c, q, k, ids, sc)


def test_empty_and_degenerate(vf, oracle):
    c, q = _data(12, 100, 64, 3, np.float32)
    with vf.DenseIndex(c) as ix:
        ids, sc = ix.search(q[:0], 5)
        assert ids.shape == (0, 5)
        ids, sc = ix.search(q, 0)
        assert ids.shape == (3, 0)
    with vf.DenseIndex(np.zeros((0, 64), np.float32)) as ix:  # empty corpus
        ids, sc = ix.search(q, 4)
        assert np.all(ids == -1) and np.all(sc == -FLT_MAX)
    # zero rows and zero queries score 0 (sklearn divides by 1)
    c2 = c.copy()
    c2[[3, 50]] = 0
    q2 = q.copy()
    q2[1] = 0
    with vf.DenseIndex(c2) as ix:
        ids, sc = ix.search(q2, 100)
    _assert_exact(oracle, c2, q2, 100, ids, sc)
    assert np.all(sc[1] == 0.0) and ids[1].tolist() == list(range(100))


# ---- golden vectors produced by the real reference ----------------------------------------------
@pytest.mark.parametrize("ci", range(len(GI.G2_CASES)))
def test_golden_step3_batch(vf, oracle, ci):
    g = load_golden(f"g2_step3_batch_case{ci}.npz")
    chunks, evid, k = GI.g2_inputs(ci)
    assert str(g["input_sha"]) == GI.sha(chunks, evid)
    from veritasfi_amd.retrieval import top_chunks_from_embeddings
    ids, sims = top_chunks_from_embeddings(evid, chunks, k)
    kk = chunks.shape[0] if k == -1 else k
    for e in range(evid.shape[0]):
        assert_topk_equiv(g["ids"][e], g["sims"][e], ids[e], sims[e])
    clear = g["min_gap"] > 1e-5
    assert np.array_equal(ids[clear], g["ids"][clear])
    _assert_exact(oracle, chunks, evid, kk, ids, sims)


def test_golden_continuous_and_ties(vf, oracle):
    g = load_golden("g1_continuous_select_top_chunks.npz")
    chunks, evid = GI.g1_inputs()
    from veritasfi_amd.retrieval import top_chunks_from_embeddings
    for k in (3, 8):
        ids, sims = top_chunks_from_embeddings(evid, chunks, k)
        assert_topk_equiv(g[f"ids_k{k}"], g["sim_row"][g[f"ids_k{k}"]], ids[0], sims[0])
    chunks, evid, groups = GI.g4_inputs()
    ids, sims = top_chunks_from_embeddings(evid, chunks, -1)
    _assert_exact(oracle, chunks, evid, chunks.shape[0], ids, sims)
    g3 = load_golden("g3_cosine_fp16_inputs.npz")
    corpus, queries = GI.g3_inputs()
    sim = vf.cosine_scores(queries, corpus.astype(np.float32))
    assert np.max(np.abs(sim - g3["sim"])) <= 1e-6
    assert np.array_equal(_bits(sim), _bits(oracle.cosine(queries, corpus.astype(np.float32))))


# ---- fused MFMA scan path -------------------------------------------------------------------------
FUSED_CASES = [
    # n, d, nq, k, dtype
    (50_000, 768, 64, 100, np.float16),   # C2 shape, scaled down
    (40_000, 768, 5, 10, np.float16),     # one 32-query tile
    (30_000, 768, 70, 100, np.float16),   # two batches (64 + 6)
    (60_000, 1024, 64, 100, np.float16),  # bge-large dim
    (33_000, 384, 17, 50, np.float16),    # 6 segments -> G = 2/3
    (25_000, 100, 8, 20, np.float16),     # d padded to 128, scan copy
    (45_000, 768, 64, 100, np.float32),   # fp32 corpus: fp16 scan copy + fp32 exact rows
    (70_000, 768, 3, 1000, np.float16),   # large k (k' = 1250)
    (20_000, 768, 2, 2048, np.float16),   # the reference's k = 2048
]


@pytest.mark.parametrize("n,d,nq,k,dtype", FUSED_CASES)
def test_fused_path_bit_exact(vf, oracle, n, d, nq, k, dtype):
    c, q = _data(21, n, d, nq, dtype)
    with vf.DenseIndex(c) as ix:
        ids, sc = ix.search(q, k)
        st = ix.stats()
    print("fused stats", (n, d, nq, k), st)
    assert st["path"] == 1
    _assert_exact(oracle, c, q, k, ids, sc)
    # on i.i.d. data the certificate should hold without repairs
    assert st["overflowed"] == 0 and st["exact_reruns"] <= max(1, nq // 16), st


def test_paths_agree_and_options(vf, oracle):
    c, q = _data(22, 40_000, 768, 33, np.float16)
    with vf.DenseIndex(c) as ix:
        a = ix.search(q, 100)
        ix.set_option("force_path", 2)  # chunked exact
        b = ix.search(q, 100)
        assert ix.stats()["path"] == 2
        ix.set_option("force_path", 1)
        for name, val in (("scan_g", 1), ("scan_g", 3), ("scan_g", 4), ("sample_rows", 4), ("sample_rows", 64),
                          ("refresh_every", 16), ("waves", 64), ("waves", 1024), ("margin", 8), ("cap", 2048)):
            ix.set_option(name, val)
            r = ix.search(q, 100)
            print("option", name, val, ix.stats())
            assert np.array_equal(r[0], a[0]) and np.array_equal(_bits(r[1]), _bits(a[1])), (name, val)
        with pytest.raises(RuntimeError):
            ix.set_option("no_such_option", 1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(_bits(a[1]), _bits(b[1]))
    _assert_exact(oracle, c, q, 100, *a)


def test_certificate_repairs_on_hostile_data(vf, oracle):
    """Duplicates straddling the k / k' boundary and a corpus sorted by score (candidate flood)
    must still come out exact: the certificate fails or the buffer overflows and the exact path repairs."""
    rng = np.random.default_rng(23)
    n, d = 30_000, 256
    q = rng.standard_normal((4, d)).astype(np.float32)
    base = rng.standard_normal((n, d)).astype(np.float32)
    # 400 verbatim copies of one vector close to query 0 -> a tie group wider than k' - k
    hot = (q[0] + 0.05 * rng.standard_normal(d)).astype(np.float32)
    base[rng.choice(n, 400, replace=False)] = hot
    c = base.astype(np.float16)
    with vf.DenseIndex(c) as ix:
        ids, sc = ix.search(q, 100)
        st = ix.stats()
    print("duplicate stats", st)
    _assert_exact(oracle, c, q, 100, ids, sc)
    assert st["uncertified"] >= 1 and st["exact_reruns"] >= 1
    # ascending-by-score order for query 1: every later row beats the threshold
    sims = oracle.cosine(q[1:2], c.astype(np.float32))[0]
    c_sorted = c[np.argsort(sims, kind="stable")]
    with vf.DenseIndex(c_sorted) as ix:
        ids, sc = ix.search(q, 100)
        st = ix.stats()
    print("sorted-corpus stats", st)
    _assert_exact(oracle, c_sorted, q, 100, ids, sc)


def test_certificate_bound_holds_on_halfway_point_query(vf, oracle):
    """tests/adversarial.py: the true best match is understated by 3.8e-4 (> 2^-12 + ..., < 2^-11 + ...) and is not
    among the re-scored rows.  The library must notice (uncertified) and repair; with the round-1 constant it returned
    a top-100 without the best row."""
    import adversarial as ADV
    c = ADV.build_case(oracle)
    assert not c["victim_rescored"] and c["ck_k"] > c["A"] + c["eps_old"]
    with vf.DenseIndex(c["corpus"]) as ix:
        ids, sc = ix.search(c["query"], 100)
        st = ix.stats()
    print("half-way query stats", st)
    assert st["path"] == 1
    _assert_exact(oracle, c["corpus"], c["query"], 100, ids, sc)
    assert ids[0, 0] == c["victim"]
    assert st["uncertified"] == 1 and st["exact_reruns"] == 1
    # the same corpus under benign queries still certifies (the wider eps costs no re-runs on ordinary data)
    q = np.random.default_rng(3).standard_normal((32, 768)).astype(np.float32)
    with vf.DenseIndex(c["corpus"]) as ix:
        ids, sc = ix.search(q, 100)
        st = ix.stats()
    _assert_exact(oracle, c["corpus"], q, 100, ids, sc)
    assert st["exact_reruns"] == 0


def test_sharding_invariance_and_merge(vf, oracle):
    """Per-shard search + merge == unsharded search, bit for bit, uneven shards."""
    import torch
    c, q = _data(24, 90_000, 768, 16, np.float16)
    k = 100
    full_i, full_s = oracle.search(c, q, k)
    bounds = [0, 20_000, 20_500, 61_000, 90_000]  # includes a small-path shard (500 rows)
    parts_i, parts_s = [], []
    qd = torch.from_numpy(q).cuda()
    for a, b in zip(bounds[:-1], bounds[1:]):
        with vf.DenseIndex(c[a:b], id_offset=a) as ix:
            i, s = ix.search_device(qd, k)
            parts_i.append(i.clone())
            parts_s.append(s.clone())
    mi, ms = vf.merge_topk_device(torch.stack(parts_i).contiguous(), torch.stack(parts_s).contiguous(), k)
    torch.cuda.synchronize()
    assert np.array_equal(mi.cpu().numpy(), full_i)
    assert np.array_equal(_bits(ms.cpu().numpy()), _bits(full_s))
    omi, oms = oracle.merge_topk(torch.stack(parts_i).cpu().numpy(), torch.stack(parts_s).cpu().numpy(), k)
    assert np.array_equal(omi, full_i) and np.array_equal(_bits(oms), _bits(full_s))
    # packed form (what ShardedRetriever all-gathers): one blob per shard, merged in place
    nq = q.shape[0]
    blobs = []
    for i, s in zip(parts_i, parts_s):
        blob, vi, vs = vf.packed_result_buffer(nq, k, i.device)
        vi.copy_(i)
        vs.copy_(s)
        blobs.append(blob)
    pi, ps = vf.merge_topk_packed_device(torch.cat(blobs), len(blobs), nq, k)
    torch.cuda.synchronize()
    assert np.array_equal(pi.cpu().numpy(), full_i) and np.array_equal(_bits(ps.cpu().numpy()), _bits(full_s))


def test_device_and_pipelined_api(vf, oracle):
    import torch
    c, q = _data(25, 48_000, 768, 64, np.float16)
    cd = torch.from_numpy(c).cuda()  # corpus already in HBM: borrowed, no copy
    qd = torch.from_numpy(q).cuda()
    with vf.DenseIndex(cd) as ix:
        i0, s0 = ix.search_device(qd, 100)
        assert ix.slots >= 2
        outs = []
        for step in range(6):  # two slots in flight
            slot = step % 2
            if step >= 2:
                ix.search_end(slot)
            outs.append(ix.search_begin(slot, qd, 100))
        ix.search_end(0)
        ix.search_end(1)
        torch.cuda.synchronize()
        for i, s in outs:
            assert torch.equal(i, i0) and torch.equal(s, s0)
        with pytest.raises(RuntimeError):
            ix.search_end(0)  # nothing pending
    _assert_exact(oracle, c, q, 100, i0.cpu().numpy(), s0.cpu().numpy())


def test_threads_share_one_index(vf, oracle):
    """The reference shares one retriever between request threads without a lock."""
    c, q = _data(26, 36_000, 768, 8, np.float16)
    want = oracle.search(c, q, 50)
    errs = []
    with vf.DenseIndex(c) as ix:
        def work(t):
            try:
                for _ in range(3):
                    ids, sc = ix.search(q[t % 8:t % 8 + 1], 50)
                    assert np.array_equal(ids[0], want[0][t % 8]) and np.array_equal(_bits(sc[0]), _bits(want[1][t % 8]))
            except Exception as e:  # noqa: BLE001
                errs.append(e)
        ths = [threading.Thread(target=work, args=(t,)) for t in range(6)]
        [t.start() for t in ths]
        [t.join() for t in ths]
    assert not errs, errs


def test_small_dense_ops(vf, oracle):
    from oracle import ref_numpy as R
    rng = np.random.default_rng(27)
    x = rng.standard_normal((155, 1024)).astype(np.float32)  # n ~ retrieved chunks (stress_test.py:153)
    m = vf.cosine_matrix(x)
    assert np.array_equal(_bits(m), _bits(oracle.cosine(x, x)))
    assert np.max(np.abs(m - R.similarity_matrix(x))) <= 1e-6
    assert np.all(np.abs(np.diag(m) - 1.0) <= 1e-6)
    rer = rng.standard_normal(155).astype(np.float32)
    rer[10] = rer[20]  # exact tie
    t = R.time_scores(rng.integers(-500, 500, 155)).astype(np.float32)
    t[10] = t[20]
    scores, order = vf.fuse_rank(rer, t)
    assert np.array_equal(_bits(scores), _bits(rer + t))
    assert np.array_equal(order, R.fuse_and_rank(rer, t))


def test_cosine_matrix_of_index_rows_by_id(vf, oracle):
    """vf_cosine_matrix_rows: the similarity matrix of retrieved chunks from the corpus rows in HBM (round-3 review, item 6) ==
    the canonical cosine matrix of those rows' values, bit for bit, for fp32 / fp16 / e4m3 corpora, with an id offset,
    duplicates in the id list, and the error paths."""
    from oracle import ref_numpy as R
    import torch
    rng = np.random.default_rng(31)
    c32 = rng.standard_normal((30_000, 768)).astype(np.float32)
    ids = rng.integers(0, 30_000, 100)
    ids[7] = ids[3]                                                   # the same chunk twice: cosine exactly as for a pair of equal rows
    for rows, vals, off in ((c32, c32, 0), (c32.astype(np.float16), c32.astype(np.float16).astype(np.float32), 1_000_000)):
        with vf.DenseIndex(rows, id_offset=off) as ix:
            m = ix.cosine_matrix_rows(ids + off)
            assert m.shape == (100, 100) and np.array_equal(_bits(m), _bits(oracle.cosine(vals[ids], vals[ids])))
            assert np.array_equal(_bits(m), _bits(vf.cosine_matrix(vals[ids])))
            assert ix.cosine_matrix_rows([]).shape == (0, 0)
            with pytest.raises(RuntimeError, match="outside the index"):
                ix.cosine_matrix_rows([off - 1])
            with pytest.raises(RuntimeError, match="outside the index"):
                ix.cosine_matrix_rows([off + 30_000])
    codes = _e4m3_codes(20_000, 256, 33)
    dec = R.decode_e4m3(codes).astype(np.float32)
    sel = rng.integers(0, 20_000, 40)
    with vf.DenseIndex.from_e4m3(codes) as ix:
        assert np.array_equal(_bits(ix.cosine_matrix_rows(sel)), _bits(oracle.cosine(dec[sel], dec[sel])))
    # the opt-in route of compute_similarity_mtx / rank_chunk equals the re-embed route when the rows ARE the chunks' embeddings
    table = rng.standard_normal((500, 128)).astype(np.float32)

    class FakeEmb:
        def embed_documents(self, texts):
            return [table[int(t)].tolist() for t in texts]

    picked = [int(i) for i in rng.integers(0, 500, 60)]
    with vf.DenseIndex(table) as ix:
        a = vf.compute_similarity_mtx(FakeEmb(), [str(i) for i in picked], as_torch=False)
        b = vf.compute_similarity_mtx(FakeEmb(), [str(i) for i in picked], as_torch=False, index=ix, row_ids=picked)
        assert np.array_equal(_bits(a), _bits(b))
        with pytest.raises(ValueError):
            vf.compute_similarity_mtx(FakeEmb(), ["1", "2"], index=ix, row_ids=[1])
    # a sharded handle (round 5): every shard normalises its own rows, the blocks meet on the home device in the caller's order --
    # the same bits as the one-device matrix; ids that straddle the shard bounds, a shard that owns none of them, duplicates
    with vf.DenseIndex(c32[:4001], device_ids=[0, 0, 0]) as grp:       # blocks [0, 1334), [1334, 2668), [2668, 4001)
        pick = np.array([4000, 0, 1333, 1334, 2667, 2668, 17, 17, 3999, 1335], np.int64)
        want = oracle.cosine(c32[pick], c32[pick])
        assert np.array_equal(_bits(grp.cosine_matrix_rows(pick)), _bits(want))
        only_last = np.array([2700, 3000, 2668], np.int64)
        assert np.array_equal(_bits(grp.cosine_matrix_rows(only_last)), _bits(oracle.cosine(c32[only_last], c32[only_last])))
        with pytest.raises(RuntimeError, match="outside the index"):
            grp.cosine_matrix_rows([1, 4001])


def test_drop_in_classes(vf, oracle):
    """FaissRetriever / compute_similarity_mtx / select_top_chunks with the reference's call shapes."""
    rng = np.random.default_rng(28)
    table = rng.standard_normal((3000, 128)).astype(np.float32)

    class FakeEmb:  # stands in for HuggingFaceEmbeddings (ragManager.py:50)
        def embed_query(self, text):
            return table[int(text)].tolist()

        def embed_documents(self, texts):
            return [table[int(t)].tolist() for t in texts]

    fr = vf.FaissRetriever(table[:2500].tolist(), FakeEmb())  # list-of-lists, as chroma returns them
    I, D = fr.invoke(["2600", "2700", "17"], 2048)  # (indices, distances), ensembleRetriever.py:66
    assert I.dtype == np.int64 and D.dtype == np.float32 and I.shape == (3, 2048)
    _assert_exact(oracle, table[:2500], table[[2600, 2700, 17]], 2048, I, D)
    assert I[2, 0] == 17 and abs(D[2, 0] - 1.0) < 1e-6
    mtx = vf.compute_similarity_mtx(FakeEmb(), [str(i) for i in range(40)])
    assert tuple(mtx.shape) == (40, 40)
    assert bool((mtx[3, [1, 2, 3]] > 0.9).any())  # vllmManager.py:476 usage
    assert np.array_equal(_bits(mtx.numpy()), _bits(oracle.cosine(table[:40], table[:40])))


def test_errors_are_reported_not_fatal(vf):
    from veritasfi_amd import _ffi
    with pytest.raises(RuntimeError, match="unknown dtype"):
        h = _ffi.vp()
        x = np.zeros((4, 16), np.uint8)
        import ctypes
        _ffi.check(_ffi.lib().vf_index_create(ctypes.byref(h), x.ctypes.data, 4, 16, 7, 0, 0), "create")
    with pytest.raises(RuntimeError):
        vf.DenseIndex(np.zeros((4, 16), np.float32), device_id=99)
    with vf.DenseIndex(np.ones((4, 16), np.float32)) as ix:
        with pytest.raises(ValueError):
            ix.search(np.ones((1, 8), np.float32), 1)
    # a corpus of up to 16384 rows is built without the fused scan's operands: forcing the fused path on it is refused (round 4: the
    # option fuzz found this running the scan on null operands -- a GPU memory fault), for a plain and for a sharded handle
    rows = np.random.default_rng(3).standard_normal((14078, 16)).astype(np.float16)
    q = np.ones((2, 16), np.float32)
    for kw in ({}, {"device_ids": [0, 0]}):
        with vf.DenseIndex(rows, **kw) as ix:
            ix.set_option("force_path", 1)
            with pytest.raises(RuntimeError, match="forced fused path"):
                ix.search(q, 5)
            ix.set_option("force_path", -1)
            assert ix.search(q, 5)[0].shape == (2, 5)


# ---- BASELINE configs[1] at full size: 1M x 768 fp16, B = 64, k = 100 ----------------------------
def test_c2_full_size_bit_exact(vf, oracle):
    n, d, nq, k = 1_000_000, 768, 64, 100
    rng = np.random.default_rng(1234)
    c = np.empty((n, d), dtype=np.float16)
    for i in range(0, n, 100_000):  # chunked to bound host memory
        c[i:i + 100_000] = rng.standard_normal((100_000, d), dtype=np.float32).astype(np.float16)
    q = np.random.default_rng(4321).standard_normal((nq, d)).astype(np.float32)
    with vf.DenseIndex(c) as ix:
        ids, sc = ix.search(q, k)
        st = ix.stats()
    print("C2 stats", st)
    assert st["path"] == 1 and st["overflowed"] == 0
    _assert_exact(oracle, c, q, k, ids, sc)


# ---- edge cases: branches the main cases do not reach -------------------------------------------------
def test_wide_rows_and_path_limits(vf, oracle):
    # d = 1536: the 64-query LDS image does not fit -> 32-query passes (NT = 1), here 40 queries = two passes
    c, q = _data(31, 20_000, 1536, 40, np.float16)
    with vf.DenseIndex(c) as ix:
        ids, sc = ix.search(q, 50)
        assert ix.stats()["path"] == 1
    _assert_exact(oracle, c, q, 50, ids, sc)
    # d = 2560 (Qwen3-Embedding-4B, step3_mul.py:384): no LDS-resident query tile -> chunked exact path
    c, q = _data(32, 17_000, 2560, 3, np.float16)
    with vf.DenseIndex(c) as ix:
        ids, sc = ix.search(q, 10)
        assert ix.stats()["path"] == 2
    _assert_exact(oracle, c, q, 10, ids, sc)
    # n just past the small-path limit, k = 1
    c, q = _data(33, 16_385, 256, 2, np.float16)
    with vf.DenseIndex(c) as ix:
        ids, sc = ix.search(q, 1)
        assert ix.stats()["path"] == 1
    _assert_exact(oracle, c, q, 1, ids, sc)


def test_fp32_rows_extreme_ranges_and_ties(vf, oracle):
    rng = np.random.default_rng(34)
    n, d = 24_000, 384
    c = rng.standard_normal((n, d)).astype(np.float32)
    c[::3] *= 1e-20           # tiny rows: the fp16 scan copy needs the per-row power-of-two scale
    c[1::3] *= 3e18           # huge rows
    c[5] = 0                  # zero row
    q = rng.standard_normal((6, d)).astype(np.float32)
    q[2] *= 1e-25
    with vf.DenseIndex(c) as ix:
        ids, sc = ix.search(q, 64)
        st = ix.stats()
    print("extreme-range stats", st)
    _assert_exact(oracle, c, q, 64, ids, sc)
    # every row identical: one giant tie group -> certificate fails for every query, exact path, ids 0..k-1
    c2 = np.tile(rng.standard_normal((1, 128)).astype(np.float16), (18_000, 1))
    q2 = rng.standard_normal((3, 128)).astype(np.float32)
    with vf.DenseIndex(c2) as ix:
        ids, sc = ix.search(q2, 20)
        st = ix.stats()
    assert st["exact_reruns"] == 3
    assert np.array_equal(ids, np.tile(np.arange(20), (3, 1)))
    _assert_exact(oracle, c2, q2, 20, ids, sc)


# ---- fp8 (OCP e4m3) corpus: BASELINE.json config [4] storage format; rows stay fp8 in HBM, converted in registers ----
def _e4m3_codes(n, d, seed):
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((n, d), generator=g) * 0.5
    x[::97] *= 40.0          # exercise the top of the range (|x| up to ~100) ...
    x[5::89] *= 2.0 ** -8    # ... and the subnormals
    return x.to(torch.float8_e4m3fn).view(torch.uint8).numpy().copy()


def test_hardware_e4m3_conversion_matches_oracle_table(vf):
    """The fused scan converts fp8 codes with v_cvt_scalef32_pk_f16_fp8; pin that instruction (all 256 codes, in
    both halves of a word) against the oracle's table, which is itself pinned against torch.float8_e4m3fn."""
    import ctypes
    from oracle import ref_numpy as R
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    L.vf_debug_cvt_e4m3.restype = ctypes.c_int
    L.vf_debug_cvt_e4m3.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
    codes = np.concatenate([np.arange(256, dtype=np.uint8), np.arange(255, -1, -1, dtype=np.uint8),
                            np.random.default_rng(0).integers(0, 256, 1003, dtype=np.uint8)])
    out = np.empty(codes.size, np.float32)
    _ffi.check(L.vf_debug_cvt_e4m3(codes.ctypes.data, out.ctypes.data, codes.size), "vf_debug_cvt_e4m3")
    want = R.decode_e4m3(codes)
    assert np.array_equal(np.isnan(out), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.array_equal(out[ok].view(np.uint32), want[ok].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("n,nq,k,d", [(3000, 5, 50, 1024), (40000, 96, 1000, 1024), (30000, 64, 100, 768), (25000, 8, 10, 200)])
def test_fp8_corpus_matches_oracle_on_decoded_rows(vf, oracle, n, nq, k, d):
    from oracle import ref_numpy as R
    codes = _e4m3_codes(n, d, 21)
    rows16 = R.decode_e4m3(codes).astype(np.float16)          # exact
    assert np.array_equal(rows16.astype(np.float32), R.decode_e4m3(codes))
    q = np.random.default_rng(22).standard_normal((nq, d)).astype(np.float32)
    want_i, want_s = oracle.search(rows16, q, k)
    ix = vf.DenseIndex.from_e4m3(codes)
    try:
        got_i, got_s = ix.search(q, k)
        assert ix.stats()["path"] == (0 if n <= 16384 else 1)      # the large cases run the fp8 scan kernel
        assert np.array_equal(got_i, want_i)
        assert np.array_equal(got_s.view(np.uint32), want_s.view(np.uint32))
        # the same index built from the decoded fp16 rows is indistinguishable
        ix16 = vf.DenseIndex(rows16)
        i16, s16 = ix16.search(q, k)
        ix16.close()
        assert np.array_equal(i16, got_i) and np.array_equal(s16.view(np.uint32), got_s.view(np.uint32))
    finally:
        ix.close()


@pytest.mark.gpu
def test_fp8_corpus_from_cuda_float8_tensor(vf, oracle):
    import torch
    from oracle import ref_numpy as R
    codes = _e4m3_codes(20000, 256, 23)
    t = torch.from_numpy(codes).cuda().view(torch.float8_e4m3fn)
    q = np.random.default_rng(24).standard_normal((7, 256)).astype(np.float32)
    ix = vf.DenseIndex(t)
    try:
        got_i, got_s = ix.search(q, 100)
    finally:
        ix.close()
    want_i, want_s = oracle.search(R.decode_e4m3(codes).astype(np.float16), q, 100)
    assert np.array_equal(got_i, want_i) and np.array_equal(got_s.view(np.uint32), want_s.view(np.uint32))


# ---- corpus file: disk -> HBM loader, shard-aware ------------------------------------------------------------
def test_index_from_corpus_file_matches_in_memory(vf, oracle, tmp_path):
    from veritasfi_amd import corpus_file as cf
    rng = np.random.default_rng(31)
    n, d, k = 50_000, 256, 100
    rows = rng.standard_normal((n, d)).astype(np.float16)
    q = rng.standard_normal((9, d)).astype(np.float32)
    p = str(tmp_path / "corpus.vfc")
    cf.write(p, rows)
    assert cf.info(p) == {"n": n, "d": d, "dtype": 1, "has_ids": False}
    want_i, want_s = oracle.search(rows, q, k)
    with vf.DenseIndex.from_file(p) as ix:
        i, s = ix.search(q, k)
    assert np.array_equal(i, want_i) and np.array_equal(_bits(s), _bits(want_s))
    # three shards read straight from the file, merged: identical to the unsharded answer
    import torch
    parts_i, parts_s = [], []
    for r in range(3):
        with vf.DenseIndex.from_file(p, rank=r, world=3) as ix:
            assert ix.id_offset == vf.shard_bounds(n, 3, r)[0]
            a, b = ix.search(q, k)
            parts_i.append(torch.from_numpy(a).cuda()); parts_s.append(torch.from_numpy(b).cuda())
    mi, ms = vf.merge_topk_device(torch.stack(parts_i).contiguous(), torch.stack(parts_s).contiguous(), k)
    assert np.array_equal(mi.cpu().numpy(), want_i) and np.array_equal(_bits(ms.cpu().numpy()), _bits(want_s))
    # fp32 rows and a small (dense-path) shard
    p32 = str(tmp_path / "c32.vfc")
    cf.write(p32, rows[:3000].astype(np.float32))
    with vf.DenseIndex.from_file(p32) as ix:
        i, s = ix.search(q, 10)
    _assert_exact(oracle, rows[:3000].astype(np.float32), q, 10, i, s)
    # damaged files are reported, not fatal
    with open(p, "r+b") as f:
        f.truncate(64 + n * d * 2 - 10)
    with pytest.raises(RuntimeError, match="truncated"):
        vf.DenseIndex.from_file(p)
    with pytest.raises(RuntimeError, match="cannot open"):
        vf.DenseIndex.from_file(str(tmp_path / "missing.vfc"))


# ---- BASELINE configs[2] at full size: the bench's own 10M x 768 fp16 corpus ------------------------------------------
def test_c3_10m_subset(vf, oracle):
    """The corpus bench.py times (same per-chunk seeds, built on the GPU), 64 queries through the fused scan; 8 of them
    are checked bit for bit against the oracle run over the host copy in 2.5M-row pieces (oracle.merge_topk joins the
    pieces, as the sharded path would).  No candidate buffer may overflow."""
    import torch
    import bench
    n, d, nq, k = 10_000_000, 768, 64, 100
    dev = torch.device("cuda", 0)
    corpus = bench.make_shard(torch, 0, n, d, dev, "f16")
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    q = torch.randn((nq, d), generator=g, device=dev, dtype=torch.float32)
    with vf.DenseIndex(corpus) as ix:
        ids, sc = ix.search_device(q, k)
        st = ix.stats()
    torch.cuda.synchronize()
    print("C3 stats", st)
    assert st["path"] == 1 and st["overflowed"] == 0 and st["n_queries"] == nq
    ids, sc = ids.cpu().numpy(), sc.cpu().numpy()
    for r in range(nq):
        assert_ranked(ids[r], sc[r])
    pick = [0, 9, 18, 27, 36, 45, 54, 63]
    qh = q[pick].cpu().numpy()
    piece = 2_500_000
    parts_i, parts_s = [], []
    for a in range(0, n, piece):
        rows = corpus[a:a + piece].cpu().numpy()          # 3.8 GB of host memory at a time
        i, s = oracle.search(rows, qh, k, id_offset=a)
        parts_i.append(i); parts_s.append(s)
        del rows
    wi, ws = oracle.merge_topk(np.stack(parts_i), np.stack(parts_s), k)
    assert np.array_equal(ids[pick], wi)
    assert np.array_equal(_bits(sc[pick]), _bits(ws))
    # BASELINE configs[2]'s partitioning at full size: the eight 1.25M-row shards (a different launch policy: CU-partitioned
    # streams, overlapping scans) merged by the product == the whole corpus, bit for bit, all 64 queries
    sh_i, sh_s = [], []
    for r in range(8):
        lo, hi = vf.shard_bounds(n, 8, r)
        with vf.DenseIndex(corpus[lo:hi], id_offset=lo) as sx:
            i, s_ = sx.search_device(q, k)
            sh_i.append(i.clone()); sh_s.append(s_.clone())
            assert sx.stats()["overflowed"] == 0
    mi, ms = vf.merge_topk_device(torch.stack(sh_i).contiguous(), torch.stack(sh_s).contiguous(), k)
    torch.cuda.synchronize()
    assert np.array_equal(mi.cpu().numpy(), ids) and np.array_equal(_bits(ms.cpu().numpy()), _bits(sc))


@pytest.mark.parametrize("n,d,nq,k,want_kernel", [
    (60_000, 768, 64, 100, 2),      # the headline shape: image 96 KB + four 12-KB rings
    (60_000, 768, 9, 10, 2),        # one N-tile
    (50_011, 100, 64, 50, 1),       # dp = 128: two segments per row, fewer than the ring is deep -> k_scan; ragged last tile
    (50_011, 200, 64, 50, 2),       # dp = 256: four segments, ragged last tile
    (40_000, 384, 40, 100, 2),      # six segments
    (40_000, 1024, 64, 100, 1),     # 128 KB image: the rings do not fit beside it -> k_scan
    (40_000, 1024, 20, 100, 2),     # ... but they do beside a 32-query image
    (30_000, 1536, 64, 10, 2),      # too wide for a 64-query image at all: 32-query passes, whose 96 KB image leaves room for the rings
    (20_000, 2400, 8, 10, 1),       # dp = 2432: a 152 KB image, register kernel only
])
def test_scan_kernels_agree_and_match_oracle(vf, oracle, n, d, nq, k, want_kernel):
    """k_scan2 (whole-line LDS-DMA corpus loads, one wave per SIMD) where image + rings + candidate stage fit the LDS, k_scan
    (register loads) elsewhere and on request: both bit-identical to the oracle; vf_search_stats names the kernel that ran;
    with and without the CU split / overlapping scans (speed options only)."""
    c, q = _data(90 + d % 7, n, d, nq, np.float16)
    want_i, want_s = oracle.search(c, q, k)
    with vf.DenseIndex(c) as ix:
        ix.set_option("force_path", 1)
        for impl, aux, ov in ((2, -1, -1), (1, -1, -1), (2, 0, 0), (2, 32, 0)):
            ix.set_option("scan_impl", impl)
            ix.set_option("overlap_scans", ov)
            i, s_ = ix.search(q, k)
            st = ix.stats()
            assert np.array_equal(i, want_i) and np.array_equal(_bits(s_), _bits(want_s)), (impl, aux, ov)
            assert st["path"] == 1 and st["exact_reruns"] == 0 and st["scan_kernel"] == (want_kernel if impl == 2 else 1), st
    with vf.DenseIndex(c) as ix:                        # the split is fixed when a slot's streams are created
        ix.set_option("force_path", 1)
        ix.set_option("aux_cus", 0)
        i, s_ = ix.search(q, k)
        assert ix.stats()["aux_cus"] == 0 and np.array_equal(i, want_i) and np.array_equal(_bits(s_), _bits(want_s))


@pytest.mark.parametrize("n,nq,k", [(60_000, 64, 100), (40_003, 64, 100), (50_000, 20, 10), (30_000, 1, 2048), (300_000, 33, 100)])
def test_scan2r_half_image_in_registers_matches_oracle_and_scan2(vf, oracle, n, nq, k):
    """k_scan2r (round 6, option scan_impl = 5): fp16 rows of 768 elements with the B fragments of a row's first six segments held in
    registers and six-segment rings -- the same products in the same order as k_scan2, so ids, score bits AND the candidate counts of
    the two kernels agree, and both equal the oracle; other widths fall back to k_scan2."""
    c, q = _data(123, n, 768, nq, np.float16)
    want_i, want_s = oracle.search(c, q, k)
    with vf.DenseIndex(c) as ix:
        ix.set_option("force_path", 1)
        got = {}
        for impl in (5, 2, 5):
            ix.set_option("scan_impl", impl)
            i, s_ = ix.search(q, k)
            st = ix.stats()
            assert st["path"] == 1 and st["exact_reruns"] == 0 and st["scan_kernel"] == impl, st
            assert np.array_equal(i, want_i) and np.array_equal(_bits(s_), _bits(want_s)), impl
            got[impl] = st["candidates"]
        assert abs(got[5] - got[2]) <= max(8, got[2] // 4), got     # (the thresholds' refresh timing differs from run to run, the filter does not)
        ix.set_option("scan_impl", 5)
        for _ in range(12):      # a timing-dependent fault (an LDS-DMA landing on fragments still being read) shows as SOME runs differing
            i, s_ = ix.search(q, k)
            assert np.array_equal(i, want_i) and np.array_equal(_bits(s_), _bits(want_s))
        # the SAMPLE pass on the same operand path (sample_impl = 1) against k_scan's (0): same sample rows, same slots -- the seeds agree to
        # the rounding of a different summation order, the results bit for bit, for every sample size and sample grid
        seen = {}
        for impl, samp, sgrid in ((0, 8, -1), (1, 8, -1), (1, 16, -1), (1, 4, 7), (1, 64, 32), (1, 4, 1), (0, 16, -1)) if k <= 100 else \
                ((0, 64, -1), (1, 64, -1), (1, 64, 5), (1, 64, 1), (0, 64, -1)):     # (a deep search needs a sample that holds k' rows)
            ix.set_option("sample_impl", impl)
            ix.set_option("sample_rows", samp)
            ix.set_option("sample_grid", sgrid)
            i, s_ = ix.search(q, k)
            st = ix.stats()
            assert st["exact_reruns"] == 0 and st["overflowed"] == 0 and np.array_equal(i, want_i) and np.array_equal(_bits(s_), _bits(want_s)), (impl, samp, sgrid, st)
            seen[(impl, samp)] = st["candidates"]
        if k <= 100:
            assert abs(seen[(1, 8)] - seen[(0, 8)]) <= max(8, seen[(0, 8)] // 4) and abs(seen[(1, 16)] - seen[(0, 16)]) <= max(8, seen[(0, 16)] // 4), seen
    c2, q2 = _data(124, 40_000, 640, 64, np.float16)
    with vf.DenseIndex(c2) as ix:
        ix.set_option("force_path", 1)
        ix.set_option("scan_impl", 5)
        i, s_ = ix.search(q2, 50)
        assert ix.stats()["scan_kernel"] == 2                        # dp = 640: not this kernel
        wi, ws = oracle.search(c2, q2, 50)
        assert np.array_equal(i, wi) and np.array_equal(_bits(s_), _bits(ws))


@pytest.mark.parametrize("n,d,nq,k", [
    (60_000, 768, 64, 100),         # S = 6 segments of 128 codes: three in registers, six-segment rings
    (60_013, 768, 20, 1000),        # one query tile, deep, ragged last tile
    (50_000, 1024, 64, 100),        # S = 8: four in registers (256 of them), five-segment rings -- the shape k_scan2 has no room for
    (50_001, 1024, 31, 10),
])
def test_scan2r_on_fp8_rows_matches_oracle(vf, oracle, n, d, nq, k):
    """k_scan2r on e4m3-resident rows (scan_impl = 5; 768 and 1024 elements): the bytes go global -> LDS as whole lines through rings of
    six / five segments, a lane converts its two 16-byte pieces per 64-element chunk in registers (exactly: every e4m3 value is an fp16
    value), and the B fragments of the first three / four segments never leave the registers.  Main scan and sample pass, bit for
    bit against the oracle on the decoded rows, repeatedly (a refill landing on fragments still being read shows as SOME runs differing)."""
    from oracle import ref_numpy as R
    codes = _e4m3_codes(n, d, 90 + d % 7)
    rows16 = R.decode_e4m3(codes).astype(np.float16)
    q = np.random.default_rng(91).standard_normal((nq, d)).astype(np.float32)
    want_i, want_s = oracle.search(rows16, q, k)
    deep = k > 100
    with vf.DenseIndex.from_e4m3(codes) as ix:
        ix.set_option("force_path", 1)
        ix.set_option("scan_impl", 5)
        if deep:
            ix.set_option("sample_rows", 64)
        cands = {}
        for simpl in (0, 1, 1, 0):
            ix.set_option("sample_impl", simpl)
            for _ in range(4):
                i, s_ = ix.search(q, k)
                st = ix.stats()
                assert st["path"] == 1 and st["scan_kernel"] == 5 and st["exact_reruns"] == 0 and st["overflowed"] == 0, (simpl, st)
                assert np.array_equal(i, want_i) and np.array_equal(_bits(s_), _bits(want_s)), simpl
            cands[simpl] = st["candidates"]
        ix.set_option("scan_impl", 1)
        ix.set_option("sample_impl", 0)
        i, s_ = ix.search(q, k)
        st = ix.stats()
        assert st["scan_kernel"] == 1 and np.array_equal(i, want_i) and np.array_equal(_bits(s_), _bits(want_s))
        assert abs(cands[1] - st["candidates"]) <= max(8, st["candidates"] // 4), (cands, st)     # same filter, different refresh timing


@pytest.mark.parametrize("n,d,nq,k", [
    (60_000, 1024, 64, 100),        # S = 16 segments: six in accumulator registers, ten in LDS (80 KB), rings of four -- the reference's own width (bge-m3)
    (50_011, 1024, 17, 1000),       # one query tile, deep, ragged
    (60_000, 512, 64, 100),         # S = 8: four in registers, rings of six
    (60_001, 384, 40, 10),          # S = 6: three in registers: a tile's segments equal the ring's depth
    (45_000, 1000, 64, 50),         # d = 1000 pads to 1024
])
def test_scan2r_other_fp16_widths_match_oracle(vf, oracle, n, d, nq, k):
    """k_scan2r's other fp16 shapes (scan_impl = 5): main scan and sample pass against the oracle, repeatedly, and against k_scan2 / k_scan."""
    c, q = _data(131 + d % 13, n, d, nq, np.float16)
    want_i, want_s = oracle.search(c, q, k)
    with vf.DenseIndex(c) as ix:
        ix.set_option("force_path", 1)
        if k > 100:
            ix.set_option("sample_rows", 64)
        for impl, simpl in ((5, 0), (5, 1), (5, 1), (4, 0), (1, 0), (5, 1)):
            ix.set_option("scan_impl", impl)
            ix.set_option("sample_impl", simpl)
            for _ in range(3):
                i, s_ = ix.search(q, k)
                st = ix.stats()
                assert st["path"] == 1 and st["exact_reruns"] == 0 and st["overflowed"] == 0, (impl, simpl, st)
                assert st["scan_kernel"] == (5 if impl == 5 else st["scan_kernel"]), (impl, st)
                assert np.array_equal(i, want_i) and np.array_equal(_bits(s_), _bits(want_s)), (impl, simpl)


@pytest.mark.parametrize("n,d,dtype,waves", [(21_845, 384, "f16", 8192), (17_000, 768, "f16", 8192), (20_011, 1024, "fp8", 8192), (16_500, 512, "f16", 1024)])
def test_scan2r_sample_pass_with_ranges_shorter_than_their_sample_part(vf, oracle, n, d, dtype, waves):
    """Ranges of ~10 rows under a sample part of 128+ (many waves over few rows, sample_rows 64): a sample tile then starts past its
    range's end -- for the last ranges past the corpus's.  The one-statement segment DMA took its scalar base from the tile's first row
    and read there (a GPU memory fault in the round-6 soak, fuzz seed 111); the base is clamped into the part now.  Rows end exactly at
    the allocation here (a fresh device buffer per index), results against the oracle."""
    from oracle import ref_numpy as R
    rng = np.random.default_rng(151)
    q = rng.standard_normal((3, d)).astype(np.float32)
    if dtype == "fp8":
        codes = _e4m3_codes(n, d, 152)
        rows = R.decode_e4m3(codes).astype(np.float16)
        make = lambda: vf.DenseIndex.from_e4m3(codes)
    else:
        rows = rng.standard_normal((n, d)).astype(np.float16)
        make = lambda: vf.DenseIndex(rows)
    want_i, want_s = oracle.search(rows, q, 500)
    with make() as ix:
        for name, val in (("force_path", 1), ("scan_impl", 5), ("sample_impl", 1), ("waves", waves), ("sample_rows", 64)):
            ix.set_option(name, val)
        for sgrid in (-1, 0, 1, 7, 1024):
            ix.set_option("sample_grid", sgrid)
            i, s_ = ix.search(q, 500)
            assert np.array_equal(i, want_i) and np.array_equal(_bits(s_), _bits(want_s)), sgrid


def test_fp16_rows_of_1024_above_1_1m_take_scan2r_by_default(vf, oracle):
    """The reference's own embedding width (bge-m3: 1024, config/example.yaml:3) as fp16 rows: above 1.1M rows the default is k_scan2r on the CU
    split (k_scan2's image does not fit this width; k_scan served it before): the rule's own path, no option set, against the oracle."""
    n, d, nq, k = 1_150_016, 1024, 8, 100
    c, q = _data(141, n, d, nq, np.float16)
    want_i, want_s = oracle.search(c, q, k)
    with vf.DenseIndex(c) as ix:
        for _ in range(3):
            i, s_ = ix.search(q, k)
            st = ix.stats()
            assert st["path"] == 1 and st["scan_kernel"] == 5 and st["exact_reruns"] == 0, st
            assert np.array_equal(i, want_i) and np.array_equal(_bits(s_), _bits(want_s))
        ix.set_option("scan_impl", 1)
        i, s_ = ix.search(q, k)
        assert ix.stats()["scan_kernel"] == 1 and np.array_equal(i, want_i) and np.array_equal(_bits(s_), _bits(want_s))


def test_e4m3_rows_above_1_1m_take_scan2r_and_its_sample_pass_by_default(vf, oracle):
    """The default for e4m3 rows of 768 / 1024 elements above 1.1M rows is k_scan2r with its own sample pass (round 6, after the filter
    rewrite: 0.694-0.698 against k_scan's 0.627-0.656 of peak at 10M x 768): the rule's own path, no option set, against the oracle."""
    from oracle import ref_numpy as R
    n, d, nq, k = 1_150_016, 768, 12, 100
    codes = _e4m3_codes(n, d, 97)
    rows16 = R.decode_e4m3(codes).astype(np.float16)
    q = np.random.default_rng(98).standard_normal((nq, d)).astype(np.float32)
    want_i, want_s = oracle.search(rows16, q, k)
    with vf.DenseIndex.from_e4m3(codes) as ix:
        for _ in range(3):
            i, s_ = ix.search(q, k)
            st = ix.stats()
            assert st["path"] == 1 and st["scan_kernel"] == 5 and st["exact_reruns"] == 0, st
            assert np.array_equal(i, want_i) and np.array_equal(_bits(s_), _bits(want_s))
        ix.set_option("scan_impl", 1)
        i, s_ = ix.search(q, k)
        assert ix.stats()["scan_kernel"] == 1 and np.array_equal(i, want_i) and np.array_equal(_bits(s_), _bits(want_s))


@pytest.mark.parametrize("n,d,nq,k,want_kernel", [
    (60_000, 768, 64, 100, 2),      # six 128-code segments per row beside a 96 KB fp16 query image
    (50_000, 1024, 24, 10, 2),      # one N-tile: 64 KB image
    (50_000, 1024, 64, 100, 1),     # 128 KB image: register kernel
    (40_003, 640, 64, 50, 2),       # five segments, ragged last tile
    (40_000, 256, 64, 50, 1),       # two segments: fewer than the ring is deep
])
def test_scan_kernels_agree_on_fp8_rows(vf, oracle, n, d, nq, k, want_kernel):
    """e4m3-resident rows through k_scan2 (option scan_impl = 3: the bytes go global -> LDS as whole lines; a lane reads two
    16-byte pieces per 64-element chunk and converts them in registers) and through k_scan (the default for e4m3 rows: it
    measured faster): both bit-identical to the oracle on the decoded rows."""
    from oracle import ref_numpy as R
    codes = _e4m3_codes(n, d, 70 + d % 11)
    rows16 = R.decode_e4m3(codes).astype(np.float16)
    q = np.random.default_rng(71).standard_normal((nq, d)).astype(np.float32)
    want_i, want_s = oracle.search(rows16, q, k)
    ix = vf.DenseIndex.from_e4m3(codes)
    try:
        ix.set_option("force_path", 1)
        for impl in (3, 2, 1):                              # 3 = k_scan2 with rows converted in registers
            ix.set_option("scan_impl", impl)
            i, s_ = ix.search(q, k)
            st = ix.stats()
            assert np.array_equal(i, want_i) and np.array_equal(_bits(s_), _bits(want_s)), impl
            want = {3: want_kernel}.get(impl, 1)
            assert st["path"] == 1 and st["exact_reruns"] == 0 and st["scan_kernel"] == want, (impl, st)
        with pytest.raises(RuntimeError):                   # (1 .. 5 exist; 4 = k_scan2 and never k_scan2r, which e4m3 rows do not take anyway)
            ix.set_option("scan_impl", 6)
    finally:
        ix.close()


@pytest.mark.parametrize("nq", [1, 3, 20])
def test_few_queries_over_a_large_corpus_stay_on_the_fused_path(vf, oracle, nq):
    """The serve path's call shape at scale (one question + up to three hyde chunks, ensembleRetriever.py:64-66) over a
    2M-row shard: the thresholds must keep rising although a workgroup stages only a few dozen candidates per query
    (round 3: publication blocks scale with the query count) -- no candidate list overflows, nothing is re-run exactly,
    and the result is the oracle's bit for bit."""
    import torch
    import bench
    n, d, k = 2_000_000, 768, 100
    corpus = bench.make_shard(torch, 0, n, d, torch.device("cuda", 0), "f16")
    q = np.random.default_rng(77).standard_normal((nq, d)).astype(np.float32)
    with vf.DenseIndex(corpus) as ix:
        ids, sc = ix.search(q, k)
        st = ix.stats()
    print("few-queries stats", nq, st)
    assert st["path"] == 1 and st["overflowed"] == 0 and st["exact_reruns"] == 0 and st["uncertified"] == 0
    assert st["max_candidates"] < 4096
    wi, ws = oracle.search(corpus.cpu().numpy(), q, k)
    assert np.array_equal(ids, wi) and np.array_equal(_bits(sc), _bits(ws))


# ---- BASELINE configs[3] at size: 5M x 768, ONE query -> top-100 -> cross-encoder over 100 pairs -> rank_chunk -> top-20 ----
def test_c4_5m_end_to_end(vf, oracle):
    """configs[3], text leg, every stage on the GPU at the configured sizes: embed_query (bge-base shape) -> exact top-100
    over the first 5M rows of the bench's corpus (ids and score bits against the oracle over the host copy, in 1.25M-row
    pieces) -> HipReranker.compute_score over 100 pairs of ~512 tokens (bge-reranker-base shape) -> rank_chunk -> at most
    20 chunks, equal to the oracle's restatement of rank_chunk fed with the same model outputs."""
    import sys
    import torch
    import bench
    from datetime import datetime
    from oracle import ref_numpy as R
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from _synth import HashTokenizer, sentence
    from bench_rerank import random_encoder
    n, d, k = 5_000_000, 768, 100
    dev = torch.device("cuda", 0)
    corpus = bench.make_shard(torch, 0, n, d, dev, "f16")
    e_enc, e_cfg = random_encoder("bert-base", head=0)
    r_enc, r_cfg = random_encoder("xlmr-base", head=1, vocab=32000)
    emb = vf.HipEmbeddings(HashTokenizer(e_cfg["vocab"]), e_enc, max_length=512, batch_size=100)
    rr = vf.HipReranker(HashTokenizer(r_cfg["vocab"]), r_enc, max_length=512)
    rng = np.random.default_rng(5)
    question = sentence(rng, 16)
    qv = np.asarray(emb.embed_query(question), np.float32)[None, :]
    assert qv.shape == (1, d) and abs(float(np.linalg.norm(qv)) - 1.0) < 1e-3
    with vf.DenseIndex(corpus) as ix:
        ids, sc = ix.search(qv, k)
        st = ix.stats()
    assert st["path"] == 1 and st["overflowed"] == 0 and st["exact_reruns"] == 0, st
    piece, parts_i, parts_s = 1_250_000, [], []
    for a in range(0, n, piece):
        rows = corpus[a:a + piece].cpu().numpy()
        i, s_ = oracle.search(rows, qv, k, id_offset=a)
        parts_i.append(i); parts_s.append(s_)
        del rows
    wi, ws = oracle.merge_topk(np.stack(parts_i), np.stack(parts_s), k)
    assert np.array_equal(ids, wi) and np.array_equal(_bits(sc), _bits(ws))
    del corpus
    passages = [sentence(rng, 470) for _ in range(64)]
    chunks = [{"page_content": passages[int(i) % 64] + f" #{int(i)}", "bundle_id": j // 2,
               "metadata": {"date_published": f"2024-{1 + j % 12:02d}-{1 + j % 28:02d}"}} for j, i in enumerate(ids[0])]
    qt = datetime(2024, 6, 15)
    got = vf.rank_chunk(chunks, question, qt, rr, emb, chunk_topk=20)
    scores = rr.compute_score([[question, c["page_content"]] for c in chunks], batch_size=8)
    assert len(scores) == 100 and np.isfinite(scores).all()
    ts = vf.time_scores(qt, [c["metadata"]["date_published"] for c in chunks])
    embs = np.asarray(emb.embed_documents([c["page_content"] for c in chunks]), np.float32)
    want = R.rank_chunk([c["bundle_id"] for c in chunks], scores, ts, embs, 20, 0.9)
    assert got == want and 0 < len(got) <= 10
    e_enc.close(); r_enc.close()


def test_c4_mixed_modality_index_text_table_figure(vf, oracle):
    """configs[3] as ONE index at the configured size: 5M x 768 rows in HBM, one id space -- text rows (the bench's corpus),
    2000 TABLE rows embedded from table-as-text by HipEmbeddings (the reference's own route for tables: every chunk goes
    through embed_documents, src/load_data.py:120-128; table-transformer is an upstream detector, not an embedder) and a
    250k-row FIGURE segment in CLIP's joint space holding 256 real ViT-L/14-geometry image embeddings.  One text query is
    embedded by BOTH towers (bge-base shape; CLIP ViT-L/14's text tower), each vector searches its own space: top-100 of
    each against the oracle over the host copy of that space's rows (ids and score bits); a table's own text and a figure's
    own image find their rows; the text hits and the best figure hits (by caption) go through the cross-encoder and
    rank_chunk keeps at most 20, equal to the oracle's restatement fed with the same model outputs."""
    import sys
    import torch
    import bench
    from datetime import datetime
    from oracle import ref_numpy as R
    from veritasfi_amd.mixed import MixedModalIndex, MixedModalRetriever, TEXT_SPACE, CLIP_SPACE
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from _synth import HashTokenizer, sentence
    from bench_rerank import random_encoder
    from bench_vision import ClipHashTokenizer, random_clip_text, random_vit
    n, d, k = 5_000_000, 768, 100
    n_fig, n_tab, n_img = 250_000, 2000, 256
    fig_lo, tab_lo = n - n_fig, n - n_fig - n_tab
    dev = torch.device("cuda", 0)
    corpus = bench.make_shard(torch, 0, n, d, dev, "f16")
    e_enc, e_cfg = random_encoder("bert-base", head=0)
    r_enc, r_cfg = random_encoder("xlmr-base", head=1, vocab=32000)
    t_enc, t_cfg = random_clip_text("vit-l-14", vocab=8192)
    v_enc, v_cfg = random_vit("vit-l-14")
    emb = vf.HipEmbeddings(HashTokenizer(e_cfg["vocab"]), e_enc, max_length=512, batch_size=100)
    cemb = vf.HipClipTextEmbeddings(ClipHashTokenizer(t_cfg["vocab"]), t_enc)
    rr = vf.HipReranker(HashTokenizer(r_cfg["vocab"]), r_enc, max_length=512)
    rng = np.random.default_rng(7)
    # tables as text: "header | header ... ; cell | cell ..." rows, embedded by the TEXT embedder
    tables = [" ; ".join(" | ".join(sentence(rng, 3).split()) for _ in range(4)) + f" table{j}" for j in range(n_tab)]
    tvec = np.asarray(emb.embed_documents(tables), np.float32)
    assert tvec.shape == (n_tab, d)
    corpus[tab_lo:fig_lo] = torch.from_numpy(tvec.astype(np.float16)).to(dev)
    # figures: seeded "images" through the vision tower; the rest of the figure segment keeps synthetic CLIP-space rows
    px = np.random.default_rng(8).standard_normal((n_img, 3, v_cfg["image"], v_cfg["image"]), dtype=np.float32)
    ivec = np.concatenate([v_enc.forward(px[i:i + 64]) for i in range(0, n_img, 64)])
    assert ivec.shape == (n_img, d) and np.isfinite(ivec).all()
    corpus[fig_lo:fig_lo + n_img] = torch.from_numpy(ivec.astype(np.float16)).to(dev)
    segments = {"text": (0, tab_lo), "table": (tab_lo, fig_lo), "figure": (fig_lo, n)}
    question = sentence(rng, 16)
    with MixedModalIndex(corpus, segments) as mix:
        assert mix.ranges == {TEXT_SPACE: (0, fig_lo), CLIP_SPACE: (fig_lo, n)}
        hits = MixedModalRetriever(mix, emb, cemb).invoke([question], k)
        (ti, ts_), (fi, fs) = hits[TEXT_SPACE], hits[CLIP_SPACE]
        assert ti.shape == fi.shape == (1, k) and ti.max() < fig_lo and fi.min() >= fig_lo       # each vector met its own space only
        assert set(mix.modality_of(ti).ravel()) <= {"text", "table"} and set(mix.modality_of(fi).ravel()) == {"figure"}
        # exactness of both searches against the oracle over the host copies (1.25M-row pieces)
        qv = {TEXT_SPACE: np.asarray([emb.embed_query(question)], np.float32), CLIP_SPACE: np.asarray([cemb.embed_query(question)], np.float32)}
        for sp, (lo, hi), (gi, gs) in ((TEXT_SPACE, (0, fig_lo), (ti, ts_)), (CLIP_SPACE, (fig_lo, n), (fi, fs))):
            pi, ps = [], []
            for a in range(lo, hi, 1_250_000):
                rows = corpus[a:min(hi, a + 1_250_000)].cpu().numpy()
                i_, s_ = oracle.search(rows, qv[sp], k, id_offset=a)
                pi.append(i_); ps.append(s_)
                del rows
            wi, ws = oracle.merge_topk(np.stack(pi), np.stack(ps), k)
            assert np.array_equal(gi, wi) and np.array_equal(_bits(gs), _bits(ws)), sp
        # a table's own text finds its row (same embedder, fp16-rounded row: cosine ~ 1); a figure's own image finds its row.
        # The probes are embedded in the batch they were ingested in: with random-init weights every table vector lies within
        # 1e-7 (cosine) of every other, so only a bit-identical embedding ranks its own row first, and the product kernels -- hence
        # the last bits -- follow the batch's tile count (mid-size batches).
        probe = [17, 34, 99]
        pvec = np.asarray(emb.embed_documents(tables[:100]), np.float32)[probe]
        pt = mix.search({TEXT_SPACE: pvec}, 5)[TEXT_SPACE]
        assert pt[0][:, 0].tolist() == [tab_lo + j for j in probe] and np.all(pt[1][:, 0] > 0.9995)
        pf = mix.search({CLIP_SPACE: ivec[[3, 100, 255]]}, 5)[CLIP_SPACE]
        assert pf[0][:, 0].tolist() == [fig_lo + 3, fig_lo + 100, fig_lo + 255] and np.all(pf[1][:, 0] > 0.9995)
    del corpus
    # re-rank: the 100 text-space hits by content, the 20 best figure hits by caption; rank_chunk keeps <= 20
    passages = [sentence(rng, 470) for _ in range(64)]
    def content(i):
        i = int(i)
        if i >= fig_lo:
            return f"figure caption {sentence(np.random.default_rng(i), 24)} #{i}"
        return tables[i - tab_lo] if i >= tab_lo else passages[i % 64] + f" #{i}"
    picked = list(ti[0]) + list(fi[0][:20])
    chunks = [{"page_content": content(i), "bundle_id": j // 2, "modality": str(mix.modality_of([i])[0]),
               "metadata": {"date_published": f"2024-{1 + j % 12:02d}-{1 + j % 28:02d}"}} for j, i in enumerate(picked)]
    qt = datetime(2024, 6, 15)
    got = vf.rank_chunk(chunks, question, qt, rr, emb, chunk_topk=20)
    scores = rr.compute_score([[question, c["page_content"]] for c in chunks], batch_size=8)
    assert len(scores) == 120 and np.isfinite(scores).all()
    tsc = vf.time_scores(qt, [c["metadata"]["date_published"] for c in chunks])
    embs = np.asarray(emb.embed_documents([c["page_content"] for c in chunks]), np.float32)
    want = R.rank_chunk([c["bundle_id"] for c in chunks], scores, tsc, embs, 20, 0.9)
    assert got == want and 0 < len(got) <= 10
    for h in (e_enc, r_enc, t_enc, v_enc):
        h.close()


# ---- BASELINE configs[4] shape: fp8-e4m3 rows, d = 1024, B = 1024 queries, k = 1000 -------------------------------------
def test_c5_shape(vf, oracle):
    from oracle import ref_numpy as R
    n, d, nq, k = 200_000, 1024, 1024, 1000
    codes = _e4m3_codes(n, d, 41)
    rows16 = R.decode_e4m3(codes).astype(np.float16)
    q = np.random.default_rng(42).standard_normal((nq, d)).astype(np.float32)
    ix = vf.DenseIndex.from_e4m3(codes)
    try:
        got_i, got_s = ix.search(q, k)
        st = ix.stats()
    finally:
        ix.close()
    print("C5 stats", st)
    assert st["path"] == 1 and st["n_queries"] == nq
    want_i, want_s = oracle.search(rows16, q, k)
    bad = np.nonzero((got_i != want_i).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} of {nq} queries differ, first {bad[:5].tolist()}"
    assert np.array_equal(got_s.view(np.uint32), want_s.view(np.uint32))


@pytest.mark.gpu
def test_c5_10m_sharding_invariance_and_subset(vf, oracle):
    """BASELINE configs[4] at FULL size on one GPU: 10M x 1024 e4m3 rows (the corpus bench.py times), 1024 queries, k = 1000
    through k_scan_wide.  (i) A size-independent property: the result over the whole corpus equals the merge of the results
    over the config's eight 1.25M-row shards, bit for bit -- the product's merge and the oracle's; (ii) four of the queries
    against the oracle run over the decoded host copy, shard by shard.  No candidate list may overflow."""
    import torch
    import bench
    n, d, nq, k, G = 10_000_000, 1024, 1024, 1000, 8
    dev = torch.device("cuda", 0)
    corpus = bench.make_shard(torch, 0, n, d, dev, "fp8")
    g = torch.Generator(device=dev)
    g.manual_seed(4321)
    q = torch.randn((nq, d), generator=g, device=dev, dtype=torch.float32)
    with vf.DenseIndex(corpus) as ix:
        ids, sc = ix.search_device(q, k)
        ids, sc = ids.clone(), sc.clone()
        st = ix.stats()
    torch.cuda.synchronize()
    print("C5 full-size stats", st)
    assert st["path"] == 1 and st["wide_launches"] >= 1 and st["overflowed"] == 0 and st["n_queries"] == nq
    pick = [0, 341, 682, 1023]
    qh = q[pick].cpu().numpy()
    parts_i, parts_s, want_i, want_s = [], [], [], []
    for r in range(G):
        lo, hi = vf.shard_bounds(n, G, r)
        with vf.DenseIndex(corpus[lo:hi], id_offset=lo) as sx:
            i, s = sx.search_device(q, k)
            parts_i.append(i.clone()); parts_s.append(s.clone())
            assert sx.stats()["overflowed"] == 0
        rows = bench.host_rows(torch, corpus, lo, hi)            # decoded e4m3 values as fp16 (exact), 2.6 GB at a time
        oi, os_ = oracle.search(rows, qh, k, id_offset=lo)
        want_i.append(oi); want_s.append(os_)
        del rows
    mi, ms = vf.merge_topk_device(torch.stack(parts_i).contiguous(), torch.stack(parts_s).contiguous(), k)
    torch.cuda.synchronize()
    ids, sc, mi, ms = ids.cpu().numpy(), sc.cpu().numpy(), mi.cpu().numpy(), ms.cpu().numpy()
    assert np.array_equal(mi, ids) and np.array_equal(_bits(ms), _bits(sc)), "eight shards merged != the whole corpus"
    omi, oms = oracle.merge_topk(torch.stack(parts_i).cpu().numpy(), torch.stack(parts_s).cpu().numpy(), k)
    assert np.array_equal(omi, ids) and np.array_equal(_bits(oms), _bits(sc))
    wi, ws = oracle.merge_topk(np.stack(want_i), np.stack(want_s), k)
    assert np.array_equal(ids[pick], wi) and np.array_equal(_bits(sc[pick]), _bits(ws))
    for r in pick:
        assert_ranked(ids[r], sc[r])


# ---- N > 1 on ONE GPU: two fresh processes, real shards, the default packed exchange ------------------------------------
_WORLD2_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["VF_ROOT"])
import numpy as np, torch, torch.distributed as dist
import veritasfi_amd as vf
from oracle import canonical as C   # checker only

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["VF_PORT"], rank=rank, world_size=world)
torch.cuda.set_device(0)
rng = np.random.default_rng(7)
corpus = rng.standard_normal((60_001, 256)).astype(np.float32).astype(np.float16)
for nq, k in ((7, 51), (64, 100)):          # nq * k odd: part strides are padded to 16 bytes
    q = np.random.default_rng(8 + nq).standard_normal((nq, 256)).astype(np.float32)
    lo, hi = vf.shard_bounds(corpus.shape[0], world, rank)
    with vf.DenseIndex(corpus[lo:hi], id_offset=lo) as ix:
        sr = vf.ShardedRetriever(ix)        # DEFAULT constructor: packed blob, one all-gather, HIP merge
        assert sr._packed and sr.world == world
        ids, sc = sr.search(torch.from_numpy(q).cuda(), k)
        torch.cuda.synchronize()
        assert ix.stats()["path"] == 1      # each shard ran the fused scan
    fi, fs = C.search(corpus, q, k)
    assert np.array_equal(ids.cpu().numpy(), fi), "ids differ from the unsharded oracle"
    assert np.array_equal(sc.cpu().numpy().view(np.uint32), fs.view(np.uint32)), "score bits differ"
# data-parallel re-rank (ShardedScorer): each rank scores its block of the pairs with a real cross-encoder replica; the
# gathered logits must be the unsharded forward's, bit for bit (a pair's logit does not depend on its batch)
sys.path.insert(0, os.path.join(os.environ["VF_ROOT"], "tools"))
from bench_rerank import random_encoder
SH = dict(vocab=500, hidden=128, layers=2, heads=2, ffn=512, max_pos=130, type_vocab=1, roberta_pad_idx=1, pooling=0,
          normalize=0, head=1, ln_eps=1e-5)
from veritasfi_amd import _ffi
import ctypes
c = _ffi.EncoderConfig(**SH); n16 = _ffi.c_i64(0); n32 = _ffi.c_i64(0)
_ffi.lib().vf_encoder_weight_sizes(ctypes.byref(c), ctypes.byref(n16), ctypes.byref(n32))
wr = np.random.default_rng(3)
enc = vf.HipEncoder(SH, (wr.standard_normal(n16.value, dtype=np.float32) * 0.05).astype(np.float16),
                    wr.standard_normal(n32.value, dtype=np.float32) * 0.05 + 0.5)
ids = np.random.default_rng(4).integers(5, 500, size=(11, 64)).astype(np.int32)
mask = np.ones_like(ids); mask[3, 40:] = 0
whole = enc.forward(ids, mask).reshape(-1)
got = vf.ShardedScorer(lambda lo, hi: enc.forward(ids[lo:hi], mask[lo:hi]).reshape(-1))(11)
enc.close()
assert got.shape == (11,) and np.array_equal(got.view(np.uint32), whole.view(np.uint32)), (got, whole)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_packed_world2(vf, tmp_path):
    """ShardedRetriever's default branch with world = 2: two child processes share the one GPU (gloo carries the blob
    through the host; RCCL refuses two ranks on one device), each holds a real DenseIndex shard, and the merged result
    must equal the unsharded oracle bit for bit on every rank."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w2.py"
    script.write_text(_WORLD2_WORKER)
    port = str(31000 + os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", VF_PORT=port, VF_ROOT=root, OMP_NUM_THREADS="8")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


# ---- ONE handle over several devices (single-process serving): vf_index_create_sharded / vf_index_group ------------------
def test_single_process_sharded_handle(vf, oracle, tmp_path):
    """The drop-in form of multi-GPU: FaissRetriever(embeddings, fn, device_ids=[...]) inside one process.  With one GPU
    on the box the shards all live on device 0 (three of them, uneven: 70001 rows), which exercises the same code:
    per-shard streams, peer copies of queries and packed results, one merge.  Bit-identical to the unsharded oracle."""
    import torch
    c, q = _data(51, 70_001, 256, 37, np.float16)
    k = 51                                                  # nq * k odd: padded part stride
    want_i, want_s = oracle.search(c, q, k)
    with vf.DenseIndex(c, device_ids=[0, 0, 0]) as ix:
        assert ix.shard_devices() == [0, 0, 0] and ix.n == 70_001
        assert ix.peer_access() == [True, True, True]       # same device: nothing to enable, nothing staged
        i, s = ix.search(q, k)                              # host buffers (FaissRetriever.invoke's call)
        st = ix.stats()
        assert np.array_equal(i, want_i) and np.array_equal(_bits(s), _bits(want_s))
        assert st["path"] == 1 and st["n_queries"] == 37 and st["candidates"] > 0
        qd = torch.from_numpy(q).cuda()
        di, ds = ix.search_device(qd, k)                    # device buffers on the home device
        torch.cuda.synchronize()
        assert np.array_equal(di.cpu().numpy(), want_i) and np.array_equal(_bits(ds.cpu().numpy()), _bits(want_s))
        outs = []
        for step in range(5):                               # two batches in flight, as bench.py --single-process keeps
            slot = step % 2
            if step >= 2:
                ix.search_end(slot)
            outs.append(ix.search_begin(slot, qd, k))
        ix.search_end(1); ix.search_end(0)
        torch.cuda.synchronize()
        for a, b in outs:
            assert torch.equal(a, di) and torch.equal(b, ds)
        with pytest.raises(RuntimeError):
            ix.search_end(0)
        with pytest.raises(RuntimeError, match="shards \\* k"):
            ix.search(q, 6000)
        ix.set_option("force_path", 2)                      # options reach every shard
        i2, s2 = ix.search(q[:3], k)
        assert ix.stats()["path"] == 2 and np.array_equal(i2, want_i[:3]) and np.array_equal(_bits(s2), _bits(want_s[:3]))
    # fp32 rows, k > rows of a shard, one shard on the small dense path, an empty trailing shard
    c32 = np.random.default_rng(52).standard_normal((5, 64)).astype(np.float32)
    q32 = np.random.default_rng(53).standard_normal((2, 64)).astype(np.float32)
    with vf.DenseIndex(c32, device_ids=[0, 0, 0, 0]) as ix:   # blocks of 2, 2, 1, 0 rows
        i, s = ix.search(q32, 8)
    _assert_exact(oracle, c32, q32, 8, i, s)
    assert (i[:, 5:] == -1).all()
    # the reference's class with the extra keyword, and the file loader
    class Emb:
        def embed_query(self, text): return q[int(text)].tolist()
    fr = vf.FaissRetriever(c.astype(np.float32).tolist()[:20_000], Emb(), device_ids=[0, 0])
    I, D = fr.invoke(["3", "5"], k=10)
    wi, ws = oracle.search(c[:20_000].astype(np.float32), q[[3, 5]], 10)
    assert np.array_equal(I, wi) and np.array_equal(_bits(D), _bits(ws))
    fr.index.close()
    from veritasfi_amd import corpus_file as cf
    p = str(tmp_path / "c.vfc")
    cf.write(p, c)
    with vf.DenseIndex.from_file(p, device_ids=[0, 0, 0, 0, 0]) as ix:
        i, s = ix.search(q, k)
    assert np.array_equal(i, want_i) and np.array_equal(_bits(s), _bits(want_s))
    # adopting device-resident shards (what bench.py --single-process builds)
    cd = torch.from_numpy(c).cuda()
    cuts = [0, 30_000, 30_100, 70_001]
    parts = [vf.DenseIndex(cd[a:b], id_offset=a) for a, b in zip(cuts[:-1], cuts[1:])]
    with vf.DenseIndex.group(parts) as ix:
        i, s = ix.search(q, k)
    assert np.array_equal(i, want_i) and np.array_equal(_bits(s), _bits(want_s))
    with pytest.raises(RuntimeError, match="contiguous"):
        a = vf.DenseIndex(cd[:100]); b = vf.DenseIndex(cd[100:300], id_offset=50)
        try:
            vf.DenseIndex.group([a, b])
        finally:
            a.close(); b.close()
    with pytest.raises(RuntimeError):
        vf.DenseIndex(c[:10], device_ids=[0, 99])


def test_sharded_handle_without_peer_access_takes_the_staged_exchange(vf, oracle):
    """Multi-GPU readiness on a one-GPU box: vf_debug_force_no_peer makes every shard of a handle built afterwards unreachable by
    peer copy, so queries and packed results travel home -> pinned host -> shard and back (streams of the owning device on either
    side, events across) -- the path a refused hipDeviceEnablePeerAccess falls to on a real node.  Same bits as the oracle through
    the host entry, the device entry and two batches in flight; the rows-by-id similarity matrix and rank_chunk(similarity_index=)
    work on the sharded handle (peer and staged) as on one device."""
    import ctypes
    from datetime import datetime
    import torch
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    L.vf_debug_force_no_peer.argtypes = [ctypes.c_int32]
    c, q = _data(71, 50_003, 256, 33, np.float16)
    k = 20
    want_i, want_s = oracle.search(c, q, k)
    prev = L.vf_debug_force_no_peer(1)
    try:
        ix = vf.DenseIndex(c, device_ids=[0, 0, 0])
    finally:
        L.vf_debug_force_no_peer(prev)
    with ix:
        assert ix.peer_access() == [False, False, False]
        for _ in range(3):                                           # the pinned buffers are reused from call to call
            i, s = ix.search(q, k)
            assert np.array_equal(i, want_i) and np.array_equal(_bits(s), _bits(want_s))
        qd = torch.from_numpy(q).cuda()
        outs = []
        for step in range(6):
            slot = step % 2
            if step >= 2:
                ix.search_end(slot)
            outs.append(ix.search_begin(slot, qd, k))
        ix.search_end(0); ix.search_end(1)
        torch.cuda.synchronize()
        for a, b in outs:
            assert np.array_equal(a.cpu().numpy(), want_i) and np.array_equal(_bits(b.cpu().numpy()), _bits(want_s))
        pick = np.array([50_002, 0, 16_667, 16_668, 33_335, 33_336, 5, 5], np.int64)
        cf = c.astype(np.float32)
        assert np.array_equal(_bits(ix.cosine_matrix_rows(pick)), _bits(oracle.cosine(cf[pick], cf[pick])))
    with vf.DenseIndex(c, device_ids=[0, 0]) as peer:                # a handle built after the hook was switched off is untouched
        assert peer.peer_access() == [True, True]

    # rank_chunk with the similarity matrix from the corpus rows, on one device and behind a sharded handle: the same selection
    rng = np.random.default_rng(72)
    table = rng.standard_normal((3000, 96)).astype(np.float32)
    table[1501] = table[10] + 1e-3 * rng.standard_normal(96).astype(np.float32)   # a near-duplicate across the shard bound

    class Emb:
        def embed_documents(self, texts): return [table[int(t)].tolist() for t in texts]

    class RR:
        def compute_score(self, pairs, batch_size=8): return [float((int(p[1]) * 7919) % 101) / 50.0 for p in pairs]

    rows = [10, 1501, 2999, 1499, 1500, 7, 2000, 42]
    chunks = [{"page_content": str(r), "bundle_id": b, "row_id": r, "metadata": {"date_published": "2024-03-%02d" % (b + 1)}}
              for b, r in enumerate(rows)]
    when = datetime(2024, 3, 9)
    base = vf.rank_chunk(chunks, "q", when, RR(), Emb(), chunk_topk=5)
    with vf.DenseIndex(table) as one, vf.DenseIndex(table, device_ids=[0, 0]) as two:
        got_one = vf.rank_chunk(chunks, "q", when, RR(), Emb(), chunk_topk=5, similarity_index=one)
        got_two = vf.rank_chunk(chunks, "q", when, RR(), Emb(), chunk_topk=5, similarity_index=two)
    assert got_one == base and got_two == base and len(base) >= 3 and not ({0, 1} <= set(base))   # the 0.9 rule dropped one of the twins


def test_small_dense_entry_points_reuse_their_workspace(vf, oracle):
    """vf_cosine_scores / _matrix / _matrix_rows / vf_fuse_rank lease one arena from a pool instead of 3 - 8 hipMalloc / hipFree pairs
    per call (they sit on the serve chain: rank_chunk makes two of them per request): after a warm-up call of each shape the
    allocation counter stands still, also with four request threads at once, and the results stay the oracle's."""
    import ctypes
    import threading
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    L.vf_debug_small_allocs.restype = ctypes.c_longlong
    rng = np.random.default_rng(73)
    x = rng.standard_normal((100, 768)).astype(np.float32)
    y = rng.standard_normal((7, 768)).astype(np.float32)
    want_m, want_s = oracle.cosine(x, x), oracle.cosine(y, x)
    a, b = rng.standard_normal(100).astype(np.float32), rng.random(100).astype(np.float32)

    def one_request():
        assert np.array_equal(_bits(vf.cosine_matrix(x)), _bits(want_m))
        assert np.array_equal(_bits(vf.cosine_scores(y, x)), _bits(want_s))
        sc, order = vf.fuse_rank(a, b)
        assert np.array_equal(sc, a + b) and sorted(order.tolist()) == list(range(100))

    one_request()
    n0 = L.vf_debug_small_allocs()
    for _ in range(20):
        one_request()
    assert L.vf_debug_small_allocs() == n0, "a steady-state call allocated device memory"
    errs = []
    def worker():
        try:
            for _ in range(10):
                one_request()
        except BaseException as e:   # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=worker) for _ in range(4)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs
    # concurrent callers each hold an arena of their own; one that pops a smaller arena than its call needs grows it once per size
    # class (three call shapes here): bounded by callers x shapes, and then steady again
    assert L.vf_debug_small_allocs() <= n0 + 12
    n1 = L.vf_debug_small_allocs()
    for _ in range(5):
        one_request()
    assert L.vf_debug_small_allocs() == n1


# ---- wide scan (k_scan_wide): more than 128 queries share one read of the shard ---------------------------------------
@pytest.mark.parametrize("n,d,nq,k,kind", [
    (60_000, 768, 130, 100, "f16"),      # one 256-query tile, padded
    (60_000, 768, 96, 100, "f16"),       # fp16 rows take the wide pass from 65 queries on (one read of the shard instead of two)
    (60_000, 1024, 96, 100, "fp8"),      # ... e4m3 rows only from 129: two narrow passes
    (90_001, 1024, 700, 10, "fp8"),      # three tiles, ragged row count
    (50_000, 768, 1024, 100, "fp8"),     # four tiles (one workgroup per CU), fp8 rows with dp = 768
    (120_000, 256, 1500, 50, "f16"),     # two passes (1024 + 476)
    (40_000, 384, 300, 20, "f16"),       # dp = 384: 6 chunks per tile
    (70_000, 1024, 256, 1000, "f16"),    # large k on the wide path
    (30_000, 640, 200, 10, "fp8"),       # fp8 with dp = 640 (not a multiple of 256): falls back to 64-query passes
])
def test_wide_scan_bit_exact(vf, oracle, n, d, nq, k, kind):
    from oracle import ref_numpy as R
    q = np.random.default_rng(62).standard_normal((nq, d)).astype(np.float32)
    if kind == "fp8":
        codes = _e4m3_codes(n, d, 61)
        rows16 = R.decode_e4m3(codes).astype(np.float16)
        ix = vf.DenseIndex.from_e4m3(codes)
    else:
        rows16 = np.random.default_rng(61).standard_normal((n, d)).astype(np.float32).astype(np.float16)
        ix = vf.DenseIndex(rows16)
    try:
        got_i, got_s = ix.search(q, k)
        st = ix.stats()
        ix.set_option("wide_sync", 0)                       # sibling workgroups in lock-step (a speed hint): identical answer
        sync_i, sync_s = ix.search(q, k)
        assert np.array_equal(sync_i, got_i) and np.array_equal(_bits(sync_s), _bits(got_s))
        ix.set_option("wide_sync", -1)
        ix.set_option("wide", 0)                            # the 64-query passes on the same handle: identical answer
        ref_i, ref_s = ix.search(q[:70], k)
        st64 = ix.stats()
    finally:
        ix.close()
    print("wide stats", (n, d, nq, k, kind), st)
    assert st["path"] == 1 and st["overflowed"] == 0
    # vf_search_stats says which kernel served the call (bench.py picks its roofline from it, not from the batch size)
    takes_wide = (d + 127) // 128 * 128 % (256 if kind == "fp8" else 128) == 0
    takes_wide = takes_wide and nq >= (129 if kind == "fp8" else 65)
    assert st["wide_launches"] == ((nq + 1023) // 1024 if takes_wide else 0) and st["wide_queries"] == (nq if takes_wide else 0)
    assert st64["wide_launches"] == 0 and st64["wide_queries"] == 0
    want_i, want_s = oracle.search(rows16, q, k)
    bad = np.nonzero((got_i != want_i).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} of {nq} queries differ, first {bad[:5].tolist()}"
    assert np.array_equal(_bits(got_s), _bits(want_s))
    assert np.array_equal(ref_i, want_i[:70]) and np.array_equal(_bits(ref_s), _bits(want_s[:70]))
    assert st["exact_reruns"] <= max(1, nq // 16), st


@pytest.mark.parametrize("n,d,nq,k", [
    (90_001, 1024, 700, 10),       # three query tiles, ragged row count
    (50_000, 768, 1024, 100),      # four tiles (one workgroup per CU), dp = 768: 12 K-tiles
    (70_000, 1024, 256, 1000),     # large k
    (300_000, 1024, 1500, 100),    # two passes (1024 + 476), several super-tiles per row group
    (40_000, 256, 200, 5),         # 113 main rows per row group: less than one super-tile (clamped rows, NaN inverse norms)
    (80_000, 512, 300, 2000),      # k close to the fused path's limit (k' = 3008, 4096-entry survivor area)
    (50_000, 2048, 150, 1),        # dp = 2048: 32 K-tiles per super-tile; k = 1
])
def test_wide_scan_fp8_matrix_instruction_bit_exact(vf, oracle, n, d, nq, k):
    """k_scan_wide8 (the default for e4m3 rows; index option wide_mfma = 0 selects the fp16 instruction): the e4m3 row bytes as the A operand of v_mfma_scale_f32_32x32x64_f8f6f4, the
    query as hi + lo e4m3 codes, each query's own quantisation residual as its certificate bound.  Same ids and score bits as
    the oracle, and as the fp16-instruction form on the same handle; repairs stay rare."""
    from oracle import ref_numpy as R
    q = np.random.default_rng(72).standard_normal((nq, d)).astype(np.float32)
    codes = _e4m3_codes(n, d, 71)
    rows16 = R.decode_e4m3(codes).astype(np.float16)
    with vf.DenseIndex.from_e4m3(codes) as ix:
        got_i, got_s = ix.search(q, k)                      # the default for e4m3 rows
        st = ix.stats()
        ix.set_option("wide_mfma", 0)
        f16_i, f16_s = ix.search(q, k)
        st16 = ix.stats()
    print("wide8 stats", (n, d, nq, k), st)
    assert st["path"] == 1 and st["scan_kernel"] == 4 and st16["scan_kernel"] == 3 and st["overflowed"] == 0
    assert st["wide_launches"] == (nq + 1023) // 1024 and st["wide_queries"] == nq
    want_i, want_s = oracle.search(rows16, q, k)
    bad = np.nonzero((got_i != want_i).any(axis=1))[0]
    assert bad.size == 0, f"{bad.size} of {nq} queries differ, first {bad[:5].tolist()}"
    assert np.array_equal(_bits(got_s), _bits(want_s))
    assert np.array_equal(f16_i, got_i) and np.array_equal(_bits(f16_s), _bits(got_s))
    assert st["exact_reruns"] <= max(1, nq // 16), st


def test_wide_scan_fp8_matrix_instruction_hostile_data(vf, oracle):
    """k_scan_wide8 under data its certificate cannot pass: 400 copies of a row close to query 0 (a tie group wider than
    k' - k), a corpus sorted by score (every tile raises the thresholds: the stage overflows into the global lists), and
    queries that are ALL the same vector (one query tile's 256 lists fill in step).  The per-query bound (the query's own
    e4m3 hi + lo residual) must flag what it cannot certify and the exact path must repair it: ids and score bits equal
    the oracle's in every case."""
    import torch
    from oracle import ref_numpy as R
    rng = np.random.default_rng(73)
    n, d, nq = 60_000, 512, 300
    q = rng.standard_normal((nq, d)).astype(np.float32)
    base = (rng.standard_normal((n, d)) * 0.5).astype(np.float32)
    base[rng.choice(n, 400, replace=False)] = (q[0] * 0.5 + 0.02 * rng.standard_normal(d)).astype(np.float32)
    codes = torch.from_numpy(base).to(torch.float8_e4m3fn).view(torch.uint8).numpy().copy()
    rows16 = R.decode_e4m3(codes).astype(np.float16)
    sims = oracle.cosine(q[1:2], rows16.astype(np.float32))[0]
    order = np.argsort(sims, kind="stable")
    for name, cd, r16, qq in (("duplicates", codes, rows16, q), ("sorted", codes[order], rows16[order], q),
                              ("one query 300 times", codes, rows16, np.repeat(q[:1], nq, axis=0))):
        with vf.DenseIndex.from_e4m3(np.ascontiguousarray(cd)) as ix:
            ids, sc = ix.search(qq, 100)
            st = ix.stats()
        print("wide8 hostile stats", name, st)
        assert st["path"] == 1 and st["scan_kernel"] == 4 and st["n_queries"] == nq
        _assert_exact(oracle, r16, qq, 100, ids, sc)


def test_wide_scan_hostile_data(vf, oracle):
    """Duplicates and a score-sorted corpus under 200 queries: stage flushes, overflow to the global lists, repairs."""
    rng = np.random.default_rng(63)
    n, d, nq = 40_000, 256, 200
    q = rng.standard_normal((nq, d)).astype(np.float32)
    base = rng.standard_normal((n, d)).astype(np.float32)
    hot = (q[0] + 0.05 * rng.standard_normal(d)).astype(np.float32)
    base[rng.choice(n, 400, replace=False)] = hot
    c = base.astype(np.float16)
    sims = oracle.cosine(q[1:2], c.astype(np.float32))[0]
    for corpus in (c, c[np.argsort(sims, kind="stable")]):
        with vf.DenseIndex(corpus) as ix:
            ids, sc = ix.search(q, 100)
            st = ix.stats()
        print("wide hostile stats", st)
        _assert_exact(oracle, corpus, q, 100, ids, sc)


# ---- differential fuzz across the dispatch boundaries ------------------------------------------------------------------------
@pytest.mark.gpu
def test_fuzz_across_dispatch_boundaries_bit_exact(vf, oracle):
    """tools/fuzz_search.py for 20 s (VF_TEST_FUZZ_SECONDS) on a fixed seed: random rows / dim / queries / k / dtype / data shape / options drawn to sit on
    the dispatch boundaries, every result compared bit for bit with the oracle (a 330-s run: profiles/r04_fuzz_seed1.log)."""
    import importlib.util, time
    spec = importlib.util.spec_from_file_location("fuzz_search", os.path.join(ROOT, "tools", "fuzz_search.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(20260404)
    t0, n_cases, fails, kernels = time.time(), 0, [], set()
    while time.time() - t0 < float(os.environ.get("VF_TEST_FUZZ_SECONDS", "20")):   # (45 s until round 6; the soak runs are tools/fuzz_search.py's)
        case = fz.draw_case(rng, 1e10)
        ok, st, why = fz.run_case(vf, oracle, case, repeat=2)
        n_cases += 1
        kernels.add((st.get("path"), st.get("scan_kernel")))
        if not ok:
            fails.append((case, why, st))
    print("fuzz:", n_cases, "cases; (path, scan kernel) seen:", sorted(kernels, key=str))
    assert not fails, fails[:3]
    assert n_cases >= 15 and len(kernels) >= 3


@pytest.mark.gpu
def test_repeat_runs_of_one_search_are_all_exact(vf, oracle):
    """tools/stress_repeat.py with 12 runs per case: the same search over and over (index built once), every run bit-equal to the
    oracle -- a timing-dependent fault shows up as SOME runs differing (that is how the narrow fp8-matrix-instruction variant of
    k_scan2 was caught and removed in round 4: 12 of 40 runs lost one row)."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("stress_repeat", os.path.join(ROOT, "tools", "stress_repeat.py"))
    sr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sr)
    old_argv = sys.argv
    try:
        sys.argv = ["stress_repeat.py", "--runs", "12"]
        assert sr.main() == 0
    finally:
        sys.argv = old_argv


@pytest.mark.gpu
def test_fp8_corpus_of_unnormalised_rows_neither_overflows_nor_flushes(oracle):
    """FaissRetriever(corpus_dtype="fp8") on rows of any scale (round-5 advisor): torch's cast to e4m3 does not saturate -- |x| > 448
    became a NaN code and the row dropped out of every result -- and tiny rows flushed to zero.  Rows are scaled per row by a power
    of two first (a cosine does not see it): rows scaled by 5000 and by 1e-6 are found like the others, and the scores are the
    canonical cosine of the stored codes."""
    import torch
    import veritasfi_amd as vf
    from oracle import ref_numpy
    rng = np.random.default_rng(11)  # (the `oracle` fixture: oracle.canonical)
    rows = rng.standard_normal((20_000, 256)).astype(np.float32)
    rows[::3] *= 5000.0
    rows[1::3] *= 1e-6
    rows[7] = 0.0

    class Emb:
        def embed_query(self, t):
            return rows[int(t)].tolist()

    fr = vf.FaissRetriever(rows, Emb(), corpus_dtype="fp8")
    probe = [0, 1, 2, 3, 4, 5, 9_999, 19_999]
    ids, sc = fr.invoke([str(i) for i in probe], 5)
    assert [int(i) for i in ids[:, 0]] == probe and np.all(sc[:, 0] > 0.995) and np.isfinite(sc).all()
    # the scores are those of the stored codes: decode what the constructor stored and ask the oracle
    peak = np.abs(rows).max(axis=1, keepdims=True).astype(np.float64)
    scaled = rows * np.exp2(np.floor(np.log2(448.0 / np.where(peak > 0, peak, 448.0)))).astype(np.float32)
    codes = torch.from_numpy(scaled).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert not ((codes & 0x7F) == 0x7F).any()
    want_ids, want_sc = oracle.search(ref_numpy.decode_e4m3(codes).astype(np.float16), rows[probe], 5)
    assert np.array_equal(ids, want_ids) and np.array_equal(_bits(sc), _bits(want_sc))
    with pytest.raises(ValueError, match="non-finite"):
        vf.FaissRetriever(np.array([[1.0, np.inf], [0.0, 1.0]], np.float32), Emb(), corpus_dtype="fp8")
    fr.index.close()


@pytest.mark.gpu
def test_deep_searches_with_few_queries_stay_on_the_fused_path(oracle):
    """The reference's own call shape -- k = 2048 with one to four queries (src/utils/ensembleRetriever.py:64-66) -- over a LARGE shard:
    a query's candidate list grows like k' (1 + ln(n / sample)), and until round 6 it was sized 4 k' (at most 16384), so k = 2048 from
    1M rows up and k = 1000 overflowed it and every such search took the exact re-run (correct, 56-72 ms at 5M rows instead of 2).
    Exact results AND no re-run."""
    import veritasfi_amd as vf
    rng = np.random.default_rng(21)
    rows = rng.standard_normal((1_000_000, 64)).astype(np.float32).astype(np.float16)
    q = rng.standard_normal((4, 64)).astype(np.float32)
    with vf.DenseIndex(rows) as ix:
        for nq, k in ((1, 1000), (4, 1000), (1, 2048), (4, 2048)):
            ids, sc = ix.search(q[:nq], k)
            st = ix.stats()
            assert st["path"] == 1 and st["overflowed"] == 0 and st["exact_reruns"] == 0, (nq, k, st)
            _assert_exact(oracle, rows, q[:nq], k, ids, sc)
