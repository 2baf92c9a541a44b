"""CPU-only checks: the C-ABI library loads and exports every symbol include/veritasfi_hip.h
declares (no compute calls without a GPU), host-side logic, and the sharded path over gloo."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from veritasfi_amd import build
    return build.build_hip()


def test_header_symbols_all_exported(built_lib):
    from veritasfi_amd import _ffi
    hdr = open(os.path.join(ROOT, "include", "veritasfi_hip.h")).read()
    declared = set(re.findall(r"\b(vf_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"vf_index", "vf_search_stats"}
    assert declared, "no declarations parsed"
    L = _ffi.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in the header but not exported"
    assert declared == set(_ffi.SIGNATURES), declared ^ set(_ffi.SIGNATURES)
    assert L.vf_version() == 100


def test_errors_without_gpu_are_codes_not_crashes(built_lib):
    """On a box without a GPU every entry point must fail with a code + message, never abort;
    on a GPU box this only checks argument validation."""
    import ctypes
    from veritasfi_amd import _ffi
    L = _ffi.lib()
    rc = L.vf_index_search(None, None, 1, 1, None, None)
    assert rc == -1 and "null handle" in _ffi.last_error()
    h = _ffi.vp()
    rc = L.vf_index_create(ctypes.byref(h), None, 5, 8, 0, 0, 0)
    assert rc == -1
    rc = L.vf_index_create(ctypes.byref(h), None, 0, 8, 7, 0, 0)
    assert rc == -1 and "dtype" in _ffi.last_error()
    assert L.vf_index_destroy(None) == 0
    with pytest.raises(RuntimeError):
        _ffi.check(rc, "probe")


def test_product_never_imports_oracle():
    """The product path must not route through the oracle (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "veritasfi_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "libvf_oracle" not in src, f
    code = ("import sys; import veritasfi_amd; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'")
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from veritasfi_amd import _ffi
    monkeypatch.setattr(_ffi, "_lib", None)
    monkeypatch.setattr(_ffi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _ffi.lib()


def test_shard_bounds_cover_rows():
    from veritasfi_amd.sharded import shard_bounds
    for n in (0, 1, 7, 8, 10_000_000, 1_250_001):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            assert all(hi - lo <= -(-n // w) for lo, hi in spans)


def test_time_scores_match_reference_formula():
    from datetime import datetime
    from veritasfi_amd.similarity import time_scores
    from oracle import ref_numpy as R
    qt = datetime(2025, 6, 1)
    dates = ["2025-06-01", "2025-05-22", "2024-06-01", "2023-01-01", "2025-07-01"]
    days = [abs((qt - datetime.strptime(d, "%Y-%m-%d")).days) for d in dates]
    assert np.allclose(time_scores(qt, dates), R.time_scores(days))


_GLOO_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["VF_ROOT"]); sys.path.insert(0, os.path.join(os.environ["VF_ROOT"], "tests"))
from veritasfi_amd.sharded import ShardedRetriever, shard_bounds
from oracle import canonical as C   # tests may use the oracle as the checker / stand-in shard

class OracleShard:   # stands in for DenseIndex on a box without a GPU
    def __init__(self, rows, off): self.rows, self.off = rows, off
    def search_device(self, q, k, out_ids=None, out_scores=None):
        i, s = C.search(self.rows, q.numpy(), k, id_offset=self.off)
        if out_ids is None:
            return torch.from_numpy(i), torch.from_numpy(s)
        out_ids.copy_(torch.from_numpy(i)); out_scores.copy_(torch.from_numpy(s))
        return out_ids, out_scores

def merge(ids, sc, k):
    i, s = C.merge_topk(ids.numpy(), sc.numpy(), k)
    return torch.from_numpy(i), torch.from_numpy(s)

def packed_merge(blob, nparts, nq, k):
    # NumPy stand-in for vf_merge_topk_packed_device: parts at the library's 16-byte padded stride
    from veritasfi_amd.index import packed_part_bytes
    raw = blob.numpy()
    stride = packed_part_bytes(nq, k)
    assert raw.size == nparts * stride and stride % 16 == 0
    ids = np.stack([raw[g * stride: g * stride + nq * k * 8].view(np.int64).reshape(nq, k) for g in range(nparts)])
    sc = np.stack([raw[g * stride + nq * k * 8: g * stride + nq * k * 12].view(np.float32).reshape(nq, k) for g in range(nparts)])
    return merge(torch.from_numpy(ids), torch.from_numpy(sc), k)

dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["VF_PORT"],
                        rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(5)
corpus = rng.standard_normal((5001, 64)).astype(np.float32).astype(np.float16)
q = torch.from_numpy(np.random.default_rng(6).standard_normal((7, 64)).astype(np.float32))
lo, hi = shard_bounds(corpus.shape[0], world, rank)
# both exchange forms: the default packed single-collective branch (NumPy merge standing in for the HIP kernel; nq*k
# odd on purpose: part strides are padded to 16 bytes) and the two-collective merge_fn branch
for K, kw in ((51, dict(packed_merge_fn=packed_merge)), (50, dict(merge_fn=merge))):
    sr = ShardedRetriever(OracleShard(corpus[lo:hi], lo), **kw)
    assert sr._packed == ("packed_merge_fn" in kw)
    ids, sc = sr.search(q, K)
    fi, fs = C.search(corpus, q.numpy(), K)
    assert np.array_equal(ids.numpy(), fi), "ids differ from unsharded"
    assert np.array_equal(sc.numpy().view(np.uint32), fs.view(np.uint32)), "scores differ from unsharded"
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_retriever_gloo(world, tmp_path):
    """N>1 path on CPU: world_size-2/3 gloo, oracle-backed shards -> identical to unsharded."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    port = str(29500 + (os.getpid() % 2000) + world)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), VF_PORT=port, VF_ROOT=ROOT,
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_corpus_file_roundtrip_on_cpu(tmp_path):
    """Writer / pure-Python reader of the .vfc format (header last, ids table, fp8 codes); no GPU library calls."""
    from veritasfi_amd import corpus_file as cf
    rng = np.random.default_rng(0)
    rows = rng.standard_normal((1000, 48)).astype(np.float16)
    ids = rng.permutation(10_000)[:1000].astype(np.int64)
    p = str(tmp_path / "c.vfc")
    with cf.CorpusWriter(p, 48, np.float16) as w:
        for i in range(0, 1000, 300):                       # the embed loop's batches (load_data.py:120-128)
            w.append(rows[i:i + 300], ids[i:i + 300])
    h = cf.read_header(p)
    assert h == {"n": 1000, "d": 48, "dtype": 1, "has_ids": True}
    assert np.array_equal(np.asarray(cf.rows_memmap(p)), rows) and np.array_equal(np.asarray(cf.external_ids(p)), ids)
    p8 = str(tmp_path / "c8.vfc")
    codes = rng.integers(0, 255, size=(10, 16), dtype=np.uint8)
    cf.write(p8, codes, e4m3=True)
    assert cf.read_header(p8)["dtype"] == 2 and cf.external_ids(p8) is None
    assert np.array_equal(np.asarray(cf.rows_memmap(p8)), codes)
    bad = tmp_path / "bad.vfc"
    bad.write_bytes(b"NOTACORPUS" + b"\0" * 100)
    with pytest.raises(ValueError):
        cf.read_header(str(bad))
    with pytest.raises(ValueError):
        with cf.CorpusWriter(str(tmp_path / "x.vfc"), 8) as w:
            w.append(np.zeros((2, 8), np.float16))
            w.append(np.zeros((2, 9), np.float16))
    assert not os.path.exists(str(tmp_path / "x.vfc")), "a writer left on an exception must not leave a valid-looking file"


def test_embed_loop_to_corpus_file(tmp_path):
    """The embed loop (load_data.py:120-128: batches of 100 through embed_documents) written to a corpus file."""
    from veritasfi_amd import corpus_file as cf

    class Emb:
        calls = []
        def embed_documents(self, texts):
            self.calls.append(len(texts))
            return [[float(len(t)), float(ord(t[0])), 1.0, -2.0] for t in texts]

    texts = [chr(65 + i % 26) * (1 + i % 7) for i in range(250)]
    p = str(tmp_path / "emb.vfc")
    seen = []
    n = cf.embed_to_file(p, texts, Emb(), batch_size=100, ids=list(range(1000, 1250)), on_batch=lambda a, b: seen.append((a, b)))
    assert n == 250 and Emb.calls == [100, 100, 50] and seen[-1] == (250, 250)
    rows = np.asarray(cf.rows_memmap(p))
    assert rows.shape == (250, 4) and rows.dtype == np.float16 and rows[3, 0] == len(texts[3]) and rows[3, 1] == ord(texts[3][0])
    assert np.array_equal(np.asarray(cf.external_ids(p)), np.arange(1000, 1250))
    with pytest.raises(ValueError):
        cf.embed_to_file(str(tmp_path / "e.vfc"), [], Emb())

    class Failing(Emb):
        def embed_documents(self, texts):
            if len(Failing.calls) >= 1:
                raise RuntimeError("encoder died")
            return super().embed_documents(texts)
    Failing.calls = []
    with pytest.raises(RuntimeError):
        cf.embed_to_file(str(tmp_path / "f.vfc"), texts, Failing(), batch_size=100)
    assert not os.path.exists(str(tmp_path / "f.vfc")), "a failed embed loop must not leave a truncated corpus file"


def test_generated_asm_block_of_the_wide_scan_is_in_sync():
    """k_scan_wide8's K-tile bodies are generated (tools/gen_w8_asm.py); the source carries the generator's output verbatim."""
    import subprocess
    src = open(os.path.join(ROOT, "veritasfi_amd", "csrc", "vf_kernels.hip")).read()
    a = src.index("#define VF8_ASM_E0_A \\")
    b = src.index("// WAVES = 8: one workgroup of 512 threads per CU, tile 256 rows x 256 queries")
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_w8_asm.py")], capture_output=True, text=True, check=True).stdout
    assert src[a:b].strip() == gen.strip()


def test_wide_scan_asm_bodies_drain_their_lds_reads():
    """Every K-tile body of k_scan_wide8 ends its second piece with s_waitcnt lgkmcnt(0) behind its last ds_read: the stage it read is
    handed to the NEXT tile's LDS-DMA after the following barrier, so the reads must have returned by then (the hazard class that
    tools/lint_lds_dma.py looks for in compiled code; in hand-written asm nothing else would put the wait there)."""
    import re
    src = open(os.path.join(ROOT, "veritasfi_amd", "csrc", "vf_kernels.hip")).read()
    a = src.index("#define VF8_ASM_E0_A \\")
    b = src.index("// WAVES = 8: one workgroup of 512 threads per CU, tile 256 rows x 256 queries")
    bodies = re.findall(r"#define (VF8_ASM_\w+_B) \\\n((?:.*\\\n)*.*\n)", src[a:b])
    assert len(bodies) == 5
    for name, body in bodies:
        ins = re.findall(r'"([^"\\]+)\\n\\t"', body)
        last_read = max(i for i, x in enumerate(ins) if x.startswith("ds_read"))
        assert any("lgkmcnt(0)" in x for x in ins[last_read + 1:]), name


def test_isa_lint_no_lds_reads_outstanding_where_lds_is_handed_to_a_dma():
    """tools/lint_lds_dma.py on the compiled kernels (hipcc cross-compiles here; about a minute): in no kernel -- the two that read
    across the hand-over point by design aside -- can a ds_read still be outstanding where an LDS-DMA is issued or a barrier gives the
    bytes back.  The source cannot promise that on its own: the compiler may move the consuming matrix instructions, and the waits in
    front of them, below a refill (round 4: it did, in a kernel variant that then lost rows intermittently)."""
    import importlib.util
    import shutil
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not available")
    spec = importlib.util.spec_from_file_location("lint_lds_dma", os.path.join(ROOT, "tools", "lint_lds_dma.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    assert lint.main() == 0


def test_isa_lint_follows_loop_back_edges():
    """The lint's walk is a data-flow over basic blocks: a ds_read left outstanding at the BOTTOM of a loop body reaches the LDS-DMA
    at its TOP through the back-edge (the ring-refill hazard; a single pass in program order reports nothing for this body), a
    state is handed to a branch target as it is AT the branch, and a drained loop stays clean."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lint_lds_dma", os.path.join(ROOT, "tools", "lint_lds_dma.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    ring = """
        s_mov_b32 s0, 0
    .LBB0_1:
        global_load_lds_dwordx4 v1, s[2:3]
        s_waitcnt vmcnt(0)
        s_barrier
        ds_read_b128 v[4:7], v2
        s_add_i32 s0, s0, 1
        s_cmp_lt_i32 s0, 8
        s_cbranch_scc1 .LBB0_1
        s_waitcnt lgkmcnt(0)
        s_endpgm
    """
    assert lint.lint_function(ring) == (1, 1)
    assert lint.lint_function(ring.replace("s_add_i32 s0, s0, 1", "s_waitcnt lgkmcnt(0)\n        s_add_i32 s0, s0, 1")) == (0, 0)
    early = """
        ds_read_b128 v[4:7], v2
        s_cbranch_scc1 .LBB0_2
        s_waitcnt lgkmcnt(0)
        s_barrier
        s_endpgm
    .LBB0_2:
        s_barrier
        s_endpgm
    """
    assert lint.lint_function(early) == (0, 1)      # only the branch target's barrier sees the read


def test_configure_is_explicit_and_warns_when_too_late(monkeypatch):
    """Importing the package leaves the environment alone (round 5 review); configure() sets GPU_MAX_HW_QUEUES once, respects the host's
    own setting, and says so when the runtime is already up."""
    import subprocess
    import sys
    import warnings
    import veritasfi_amd as vf
    code = ("import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import veritasfi_amd as vf; a = os.environ.get('GPU_MAX_HW_QUEUES'); "
            "r = vf.configure(); print(a, r, os.environ.get('GPU_MAX_HW_QUEUES'))")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, check=True).stdout.split()
    assert out == ["None", "True", "8"]
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "3")
    assert vf.configure() is True and os.environ["GPU_MAX_HW_QUEUES"] == "3"          # the host's setting wins
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    from veritasfi_amd import _ffi
    monkeypatch.setattr(_ffi, "loaded", lambda: True)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert vf.configure() is False and "GPU_MAX_HW_QUEUES" not in os.environ
    assert any("already initialised" in str(x.message) for x in w)
