import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

import veritasfi_amd  # noqa: E402

veritasfi_amd.configure(warn=False)   # the suite opens many handles in one process: eight hardware queues, set before any GPU call


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# ---- tests that drive single kernels through vf_debug_* hooks, or ask for a measured-and-rejected variant -------------------------
# The product library exports the C ABI of include/veritasfi_hip.h and two hooks (veritasfi_amd/build.py: KEPT_HOOKS); every other
# hook and the rejected kernel variants live in libvf_test.so (-DVF_EXPERIMENTS).  A test function that -- itself or through a
# helper of its module -- names such a hook is a HOOK TEST: skipped when the loaded library has no hooks, and the only kind that runs
# when VF_HOOK_TESTS_ONLY=1 (tests/test_gpu_hooks.py starts that run in a child process bound to libvf_test.so).
KEPT_HOOKS = ("vf_debug_force_no_peer", "vf_debug_small_allocs")
_HOOK_WORDS = ("wide8_waves", "VF_EXPERIMENTS")
_hook_cache = {}


def _names_a_hook(src: str) -> bool:
    import re
    return any(h not in KEPT_HOOKS for h in re.findall(r"vf_debug_[a-z0-9_]+", src)) or any(w in src for w in _HOOK_WORDS)


def _hook_functions(module) -> set:
    """Names of the module's functions that reach a hook: direct mention, or a call of a module function that does (fixpoint)."""
    import inspect
    import re
    key = getattr(module, "__name__", id(module))
    if key in _hook_cache:
        return _hook_cache[key]
    srcs = {}
    for name, fn in vars(module).items():
        if inspect.isfunction(fn) and getattr(fn, "__module__", None) == module.__name__:
            try:
                srcs[name] = inspect.getsource(fn)
            except (OSError, TypeError):
                pass
    hooked = {n for n, src in srcs.items() if _names_a_hook(src)}
    grew = True
    while grew:
        grew = False
        for n, src in srcs.items():
            if n not in hooked and any(re.search(r"\b" + re.escape(h) + r"\s*\(", src) for h in hooked):
                hooked.add(n)
                grew = True
    _hook_cache[key] = hooked
    return hooked


def _library_has_hooks() -> bool:
    try:
        from veritasfi_amd import _ffi
        return hasattr(_ffi.lib(), "vf_debug_gemm")
    except Exception:  # noqa: BLE001 -- no library at all: the tests that need it fail on their own
        return False


def pytest_collection_modifyitems(config, items):
    only = os.environ.get("VF_HOOK_TESTS_ONLY") == "1"
    have = None
    keep, drop = [], []
    for item in items:
        fn = getattr(item, "function", None)
        is_hook = fn is not None and fn.__name__ in _hook_functions(item.module)
        if only and not is_hook:
            drop.append(item)
            continue
        keep.append(item)
        if is_hook and item.get_closest_marker("gpu") is not None:
            if have is None:
                have = _library_has_hooks()
            if not have:
                item.add_marker(pytest.mark.skip(reason="drives a vf_debug_* hook / a rejected variant: runs against libvf_test.so in "
                                                        "tests/test_gpu_hooks.py's child process"))
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


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
