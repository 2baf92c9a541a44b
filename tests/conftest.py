import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def assert_topk_equiv(ref_ids, ref_scores, ids, scores, tol=2e-6):
    """Compare one ranked row against the REFERENCE's (BLAS-ordered) result.

    Scores must match position by position within `tol`.  Ids must match exactly except inside
    near-tie groups: a differing id at rank i is accepted only if the reference holds that id at a
    rank whose reference score is within `tol` of rank i's, or -- at the tail -- if rank i is within
    `tol` of the reference's last kept score (the swap partner fell just outside the cut).
    """
    ref_ids = np.asarray(ref_ids)
    ref_scores = np.asarray(ref_scores, dtype=np.float64)
    ids = np.asarray(ids)
    scores = np.asarray(scores, dtype=np.float64)
    assert ids.shape == ref_ids.shape and scores.shape == ref_scores.shape
    assert np.all(np.abs(scores - ref_scores) <= tol), float(np.max(np.abs(scores - ref_scores)))
    assert len(set(ids.tolist())) == len(ids), "duplicate ids in result"
    pos = {int(r): i for i, r in enumerate(ref_ids)}
    for i in np.nonzero(ids != ref_ids)[0]:
        j = pos.get(int(ids[i]))
        if j is not None:
            assert abs(ref_scores[j] - ref_scores[i]) <= tol, (i, j, ref_scores[i], ref_scores[j])
        else:
            assert abs(ref_scores[i] - ref_scores[-1]) <= tol, (i, ref_scores[i], ref_scores[-1])


def assert_ranked(ids, scores):
    """Descending scores, ties broken by lower id first (the product's declared order)."""
    ids = np.asarray(ids)
    scores = np.asarray(scores)
    valid = ids >= 0
    s, i = scores[valid], ids[valid]
    assert np.all(s[:-1] >= s[1:])
    eq = s[:-1] == s[1:]
    assert np.all(i[:-1][eq] < i[1:][eq])


@pytest.fixture(scope="session")
def oracle():
    from oracle import canonical
    canonical.build()
    return canonical
