"""Pin the CPU oracle against the REAL reference's outputs (tests/golden/, tools/gen_golden.py).

CPU-only.  Covers both oracle layers: the NumPy restatement of the literal reference path and the
canonical-order C restatement the HIP path is compared with bit-for-bit.
"""
import math

import numpy as np
import pytest

import golden_inputs as GI
from conftest import assert_ranked, assert_topk_equiv, load_golden
from oracle import ref_numpy as R


def _check_sha(g, *arrays):
    assert str(g["input_sha"]) == GI.sha(*arrays), "golden inputs drifted (NumPy generator changed?)"


def test_dot16_matches_python_emulation(oracle):
    """The canonical primitive is exactly the 16-way interleaved fmaf chain + fixed tree."""
    rng = np.random.default_rng(3)
    for d in (1, 7, 16, 17, 64, 100, 768, 1024):
        a = rng.standard_normal(d).astype(np.float32)
        b = rng.standard_normal(d).astype(np.float32)
        acc = [np.float32(0)] * 16
        for j in range(d):
            # exact product+sum in float64 is not an fma in general; use exact rational rounding
            # through python's math.fma when present, else float64 (exact for fp32 products: 48 bits)
            prod = float(a[j]) * float(b[j])  # exact: 24+24 bits fit in 53
            acc[j % 16] = np.float32(prod + float(acc[j % 16]))  # one rounding 53->24: double rounding
        # double rounding can differ from a true fma in rare half-way cases; compare with tolerance 1 ulp
        for s in (8, 4, 2):
            for l in range(s):
                acc[l] = np.float32(acc[l] + acc[l + s])
        emu = np.float32(acc[0] + acc[1])
        got = oracle.dot16(a, b)
        assert abs(float(emu) - float(got)) <= 2 * np.spacing(np.float32(abs(emu))) + 1e-30


@pytest.mark.skipif(not hasattr(math, "fma"), reason="math.fma needs Python >= 3.13")
def test_dot16_exact_fma_emulation(oracle):
    rng = np.random.default_rng(4)
    a = rng.standard_normal(768).astype(np.float32)
    b = rng.standard_normal(768).astype(np.float32)
    acc = [0.0] * 16
    for j in range(768):
        acc[j % 16] = float(np.float32(math.fma(float(a[j]), float(b[j]), acc[j % 16])))
    for s in (8, 4, 2):
        for l in range(s):
            acc[l] = float(np.float32(acc[l] + acc[l + s]))
    assert np.float32(acc[0] + acc[1]) == oracle.dot16(a, b)


def test_g1_continuous_select_top_chunks(oracle):
    g = load_golden("g1_continuous_select_top_chunks.npz")
    chunks, evid = GI.g1_inputs()
    _check_sha(g, chunks, evid)
    sim_ref = g["sim_row"]
    # numpy restatement vs the reference's sklearn call
    assert np.max(np.abs(R.cosine_similarity(evid, chunks)[0] - sim_ref)) <= 1e-6
    for k in (3, 8):
        ref_ids = g[f"ids_k{k}"]
        ids, sc = oracle.search(chunks, evid, k)
        assert_topk_equiv(ref_ids, sim_ref[ref_ids], ids[0], sc[0])
        # the NumPy restatement of the reference's own argsort line reproduces the reference's ids on its own scores
        assert np.array_equal(R.argsort_topk(sim_ref, k), ref_ids)
        assert_ranked(ids[0], sc[0])


@pytest.mark.parametrize("ci", range(len(GI.G2_CASES)))
def test_g2_step3_select_top_chunks_batch(oracle, ci):
    g = load_golden(f"g2_step3_batch_case{ci}.npz")
    chunks, evid, k = GI.g2_inputs(ci)
    _check_sha(g, chunks, evid)
    assert int(g["k"]) == k
    kk = chunks.shape[0] if k == -1 else k
    ids, sc = oracle.search(chunks, evid, kk)
    rn = R.select_top_chunks_batch(evid, chunks, k)
    for e in range(evid.shape[0]):
        assert_topk_equiv(g["ids"][e], g["sims"][e], ids[e], sc[e])
        assert_topk_equiv(g["ids"][e], g["sims"][e], rn[e][0], rn[e][1])
        assert_ranked(ids[e], sc[e])
    # where the reference itself is far from any tie the ids must match exactly
    clear = g["min_gap"] > 1e-5
    assert np.array_equal(ids[clear], g["ids"][clear])


def test_g3_cosine_matrix_fp16_inputs(oracle):
    g = load_golden("g3_cosine_fp16_inputs.npz")
    corpus, queries = GI.g3_inputs()
    _check_sha(g, corpus, queries)
    sim = oracle.cosine(queries, corpus.astype(np.float32))
    assert sim.dtype == np.float32
    assert np.max(np.abs(sim - g["sim"])) <= 1e-6
    assert np.max(np.abs(R.cosine_similarity(queries, corpus.astype(np.float32)) - g["sim"])) <= 1e-6
    # the fp16 entry point sees the same values as fp32-of-fp16 -> identical bits
    i16, s16 = oracle.search(corpus, queries, 100)
    i32, s32 = oracle.search(corpus.astype(np.float32), queries, 100)
    assert np.array_equal(i16, i32) and np.array_equal(s16.view(np.uint32), s32.view(np.uint32))
    # search == dense matrix + per-row top-k, bit for bit
    for q in range(queries.shape[0]):
        ids, sc = oracle.topk_row(sim[q], 100)
        assert np.array_equal(ids, i32[q]) and np.array_equal(sc, s32[q])


def test_g4_ties_characterisation(oracle):
    g = load_golden("g4_ties.npz")
    chunks, evid, groups = GI.g4_inputs()
    _check_sha(g, chunks, evid)
    n = chunks.shape[0]
    ids, sc = oracle.search(chunks, evid, n)
    # same multiset of ids, same scores to fp32 rounding
    assert sorted(ids[0].tolist()) == list(range(n))
    assert np.max(np.abs(np.sort(sc[0]) - np.sort(g["sims"]))) <= 1e-6
    # exact duplicates score identically in the canonical oracle and come out lower-id first
    pos = {int(r): i for i, r in enumerate(ids[0])}
    for grp in groups:
        dup = [grp[0], grp[1]]  # row and its verbatim copy
        assert sc[0][pos[dup[0]]] == sc[0][pos[dup[1]]]
        assert pos[dup[0]] + 1 == pos[dup[1]]
    assert_ranked(ids[0], sc[0])
    # zero row: sklearn divides by 1 -> cosine 0 (step3_mul.py:275 via normalize)
    assert sc[0][pos[n - 1]] == 0.0
    # the reference keeps the same tie groups adjacent, in an order we do not pin
    rpos = {int(r): i for i, r in enumerate(g["ids"])}
    assert abs(rpos[5] - rpos[40]) <= 2 and abs(rpos[10] - rpos[41]) == 1


def test_k_larger_than_n_pads(oracle):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((7, 32)).astype(np.float32)
    q = rng.standard_normal((2, 32)).astype(np.float32)
    ids, sc = oracle.search(x, q, 10)
    assert np.all(ids[:, 7:] == -1) and np.all(sc[:, 7:] == -np.finfo(np.float32).max)
    assert sorted(ids[0, :7].tolist()) == list(range(7))
    ri, rs = R.faiss_flat_ip_search(x, q, 10)
    assert np.array_equal(ri, ids) and np.max(np.abs(rs[:, :7] - sc[:, :7])) <= 1e-6


def test_merge_equals_unsharded(oracle):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((3001, 96)).astype(np.float32)
    q = rng.standard_normal((5, 96)).astype(np.float32)
    k = 50
    full_i, full_s = oracle.search(x, q, k)
    bounds = [0, 700, 1500, 1501, 3001]
    parts = [oracle.search(x[a:b], q, k, id_offset=a) for a, b in zip(bounds[:-1], bounds[1:])]
    mi, ms = oracle.merge_topk(np.stack([p[0] for p in parts]), np.stack([p[1] for p in parts]), k)
    assert np.array_equal(mi, full_i) and np.array_equal(ms.view(np.uint32), full_s.view(np.uint32))


def test_rank_fusion_restatement():
    """vllmManager.rank_chunk :443-457 -- time score + descending order."""
    t = R.time_scores([0, 10, 365, 400, -30])
    assert np.allclose(t, [1.0, 1 - 10 / 365, 0.0, 0.0, 1 - 30 / 365])
    order = R.fuse_and_rank([0.5, 2.0, 2.0, -1.0, 0.1], t)
    assert order.tolist() == [1, 2, 0, 4, 3]


def test_rank_chunk_restatement_logic():
    """vllmManager.rank_chunk :430-483 on hand-made inputs: bundle size cap, bundle-id-as-column quirk, reverse order."""
    rng = np.random.default_rng(9)
    emb = rng.standard_normal((6, 16)).astype(np.float32)
    emb[4] = emb[0]                                  # chunk 4 duplicates chunk 0
    bundles = [0, 0, 1, 2, 3, 3]
    scores = [5.0, 1.0, 4.0, 3.0, 2.0, 0.5]
    t = [0.0] * 6
    # ranked: 0,2,3,4,1,5 -> bundle 0 (size 2), bundle 1, bundle 2, then bundle 3 (chunk 4): sim[4, [0,1,2]]:
    # column 0 is chunk 0 == chunk 4 -> similarity 1 > 0.9 -> skipped; chunk 5 (bundle 3): sim[5,[0,1,2]] small -> kept
    out = R.rank_chunk(bundles, scores, t, emb, chunk_topk=10, similar_threshhold=0.9)
    assert out == [3, 2, 1, 0]
    assert R.rank_chunk(bundles, scores, t, emb, chunk_topk=3, similar_threshhold=0.9) == [1, 0]
    assert R.rank_chunk([], [], [], np.zeros((0, 16), np.float32), 5) == []


def test_e4m3_decode_matches_torch():
    """The oracle's fp8 decoder against an independent implementation (torch.float8_e4m3fn), all 256 codes;
    every finite value survives a round trip through fp16 (the product decodes fp8 rows to fp16)."""
    torch = pytest.importorskip("torch")
    if not hasattr(torch, "float8_e4m3fn"):
        pytest.skip("torch without float8_e4m3fn")
    codes = np.arange(256, dtype=np.uint8)
    ref = torch.from_numpy(codes.copy()).view(torch.float8_e4m3fn).float().numpy()
    got = R.decode_e4m3(codes)
    assert np.array_equal(np.isnan(ref), np.isnan(got))
    fin = ~np.isnan(ref)
    assert np.array_equal(ref[fin].view(np.uint32), got[fin].view(np.uint32))
    assert np.array_equal(got[fin].astype(np.float16).astype(np.float32), got[fin])
    assert np.nanmax(np.abs(got)) == 448.0 and got[1] == 2.0 ** -9


def test_oracle_under_address_and_ub_sanitizers(tmp_path):
    """oracle/vf_oracle.c built with -fsanitize=address,undefined and driven over ragged / empty / k > n / fp16 inputs
    (oracle/sanitize_check.c): the sanitizers abort on any finding."""
    import os, shutil, subprocess
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "sanitize_check")
    cmd = ["gcc", "-O1", "-g", "-mavx2", "-mfma", "-mf16c", "-ffp-contract=off", "-fopenmp", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=all", "-o", exe, os.path.join(src, "sanitize_check.c"), os.path.join(src, "vf_oracle.c"), "-lm"]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in (b.stderr or "").lower():
        pytest.skip("toolchain without sanitizer runtimes")
    assert b.returncode == 0, b.stderr
    r = subprocess.run([exe], capture_output=True, text=True, env={**os.environ, "OMP_NUM_THREADS": "4",
                                                                    "ASAN_OPTIONS": "detect_leaks=1"}, timeout=300)
    assert r.returncode == 0 and "sanitize_check ok" in r.stdout, (r.stdout, r.stderr[-2000:])
