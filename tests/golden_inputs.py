"""Deterministic inputs for the golden fixtures in tests/golden/.

The fixtures store only (seed parameters, sha256 of the generated input bytes, the REFERENCE's
outputs); the inputs are regenerated here and their checksum is verified, so a drift in NumPy's
generator fails loudly instead of silently comparing different data.  Used by tools/gen_golden.py
(which runs the real reference on these arrays) and by the tests.  No reference code here.
"""
import hashlib

import numpy as np

# (E evidences, C chunks, d, k) -- k == -1 means "all, sorted" (step3_mul.py:245-246)
G2_CASES = [
    (3, 500, 64, 5),
    (8, 2000, 768, 100),
    (64, 20000, 768, 100),
    (5, 700, 1024, -1),
    (16, 5000, 1024, 5),
    (4, 3000, 384, 10),
    (1, 150, 768, 3),
]


def sha(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def g1_inputs():
    rng = np.random.default_rng(0)
    chunks = rng.standard_normal((1000, 768)).astype(np.float32)
    evid = rng.standard_normal((1, 768)).astype(np.float32)
    return chunks, evid


def g2_inputs(ci: int):
    E, C, d, k = G2_CASES[ci]
    rng = np.random.default_rng(100 + ci)
    chunks = rng.standard_normal((C, d)).astype(np.float32)
    evid = rng.standard_normal((E, d)).astype(np.float32)
    return chunks, evid, k


def g3_inputs():
    """C2-like distribution: N(0,1) corpus rounded to fp16, fp32 queries."""
    corpus = np.random.default_rng(1234).standard_normal((4096, 768)).astype(np.float32).astype(np.float16)
    queries = np.random.default_rng(4321).standard_normal((16, 768)).astype(np.float32)
    return corpus, queries


def g4_inputs():
    """Exact ties: duplicated rows, a positively scaled copy, a zero row."""
    rng = np.random.default_rng(7)
    base = rng.standard_normal((40, 64)).astype(np.float32)
    chunks = np.vstack([base, base[[5, 10, 20]], 2.0 * base[[5]], np.zeros((1, 64), np.float32)])
    evid = base[[5]] + 0.01 * rng.standard_normal((1, 64)).astype(np.float32)
    tie_groups = [[5, 40, 43], [10, 41], [20, 42]]
    return chunks, evid, tie_groups
