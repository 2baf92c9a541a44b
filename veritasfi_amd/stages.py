"""Optional stage timing hooks with the reference profiler's names.

The reference brackets its retrieval and re-rank stages with ``src/utils/profiler.py``'s ``Profiler`` -- ``"retrieve"``
(``ensembleRetriever.py:50``), ``"retrieve_faiss"`` (``:63,135``), ``"retrieve_faiss_ts"`` (``:138,185``), ``"retrieve_bm25"``
(``:188,229``), the metric ``"retrieved_chunks"`` (``:231``) and ``"rerank"`` (``vllmChatService.py:31``) -- and reads medians /
p95 out of ``profiler.profile_data`` afterwards.  The drop-in classes keep those brackets so a p50 taken through the unchanged
orchestration has something to read::

    from utils.profiler import profiler           # the host application's own singleton
    import veritasfi_amd
    veritasfi_amd.set_profiler(profiler)          # any object with start(name) / end(name) [/ add_metric(name, value)]

Off by default: without ``set_profiler`` every bracket is a no-op (one attribute test).  ``skip=("rerank",)`` leaves a stage to
the host when its own decorator still wraps the caller (``get_rag_content`` keeps its ``@profile_function(name="rerank")`` when only
``rank_chunk`` is replaced).  ``StageTimer`` is a minimal recorder with the same interface for use without the host application.
"""
from __future__ import annotations

import contextlib
import statistics
import threading
import time

_profiler = None
_skip = frozenset()


def set_profiler(profiler, skip=()):
    """Route the stage brackets to ``profiler`` (``None`` switches them off).  Returns the previous one."""
    global _profiler, _skip
    if profiler is not None and not (hasattr(profiler, "start") and hasattr(profiler, "end")):
        raise TypeError("set_profiler: the object needs start(name) and end(name) (src/utils/profiler.py:54-85)")
    prev, _profiler, _skip = _profiler, profiler, frozenset(skip)
    return prev


def get_profiler():
    return _profiler


@contextlib.contextmanager
def stage(name: str):
    p = _profiler
    if p is None or name in _skip:
        yield
        return
    p.start(name)
    try:
        yield
    finally:
        p.end(name)


def metric(name: str, value):
    p = _profiler
    if p is not None and name not in _skip and hasattr(p, "add_metric"):
        p.add_metric(name, value)
    return value


class StageTimer:
    """``start / end / add_metric`` and ``profile_data`` shaped like the reference's ``Profiler`` (``profiler.py:10-93``); timers are
    per thread, so concurrent requests do not end each other's stages (the reference keeps one ``_active_timers`` dict)."""

    def __init__(self):
        self.profile_data, self.metrics = {}, {}
        self._lock, self._local = threading.Lock(), threading.local()

    def start(self, name):
        timers = self._local.__dict__.setdefault("timers", {})
        timers[name] = time.perf_counter()
        return name

    def end(self, name):
        t0 = self._local.__dict__.setdefault("timers", {}).pop(name, None)
        if t0 is None:
            return None
        dt = time.perf_counter() - t0
        with self._lock:
            d = self.profile_data.setdefault(name, {"calls": 0, "total_time": 0.0, "execution_times": []})
            d["calls"] += 1
            d["total_time"] += dt
            d["execution_times"].append(dt)
        return dt

    def add_metric(self, name, value):
        with self._lock:
            self.metrics.setdefault(name, []).append(value)
        return value

    def summary(self):
        """{stage: {calls, mean_ms, p50_ms, max_ms}}"""
        out = {}
        with self._lock:
            for name, d in self.profile_data.items():
                ts = d["execution_times"]
                if ts:
                    out[name] = {"calls": d["calls"], "mean_ms": 1e3 * statistics.fmean(ts), "p50_ms": 1e3 * statistics.median(ts),
                                 "max_ms": 1e3 * max(ts)}
        return out
