"""Hostile input for the fused scan's exactness certificate.

The approximate pass scores fp16(qn) . row; the certificate needs |approx - canonical| <= eps for EVERY row.  Rounding
the normalised query to fp16 moves an element by up to 2^-11 of its magnitude (11-bit significand), so the dot product
moves by up to 2^-11 -- not 2^-12.  This module builds the data on which the difference matters:

* a query whose normalised entries sit 3e-5 (relative) beside fp16 half-way points, half of them just below (they
  round DOWN) and half just above (they round UP): the rounding error vector `delta` is then orthogonal to the query and
  has norm ~3.9e-4;
* one VICTIM row = rho * q - sqrt(1 - rho^2) * delta_hat: canonical score rho, approximate score rho - 3.8e-4;
* k STRONG rows, neutral to delta, with canonical scores a few 1e-5 BELOW the victim's;
* FILLER rows, neutral to delta, whose scores sit just above the victim's APPROXIMATE score.

The top-k' by approximate score is then {strong rows, fillers}; the victim is not re-scored although it is the true
best match, and the k-th re-scored canonical score clears "k'-th approximate score + eps" for eps = 2^-12 + ... (the
round-1 constant) but not for eps = 2^-11 + ...  Everything is evaluated on the rows as they are stored (fp16), with
the oracle's canonical scores, so the construction does not depend on how the rows rounded.
"""
from __future__ import annotations

import numpy as np


def eps_bound(d: int, u16: float, fp32_corpus: bool = False) -> float:
    """The library's certificate bound (vf_api.hip make_plan) for a given fp16 unit round-off."""
    return u16 * (2.0 if fp32_corpus else 1.0) + np.sqrt(d) * 2.0 ** -24 + 2.0 * d * 2.0 ** -24 + 1e-6


def halfway_query(d: int = 768, off: float = 3e-5, seed: int = 0):
    """Unit-norm fp32 query whose entries sit `off` (relative) below (even j) / above (odd j) fp16 half-way points
    2^-5 * (1 + (2m+1) 2^-11), m in {157, 158} mixed so that the norm is 1 to ~1e-6.  Returns (q, sign) with
    sign[j] = -1 where the entry rounds down, +1 where it rounds up."""
    assert d == 768, "the m values are tuned for d = 768"
    rng = np.random.default_rng(seed)
    mu = {m: 1.0 + (2 * m + 1) * 2.0 ** -11 for m in (157, 158)}
    target = 1024.0 / d                                  # mean mantissa^2 for unit norm
    frac158 = (target - mu[157] ** 2) / (mu[158] ** 2 - mu[157] ** 2)
    n158 = int(round(frac158 * d))
    m = np.array([158] * n158 + [157] * (d - n158))
    rng.shuffle(m)
    sign = np.where(np.arange(d) % 2 == 0, -1.0, 1.0)    # -1: just below the half-way point (rounds down)
    t = 2.0 ** -5 * (1.0 + (2 * m + 1) * 2.0 ** -11) * (1.0 + sign * off)
    t = t * rng.choice([-1.0, 1.0], size=d)              # random element signs: rounding is symmetric
    return t.astype(np.float32), sign


def approx_scores(qn: np.ndarray, rows16: np.ndarray) -> np.ndarray:
    """Emulation of the scan's approximate score: fp16(qn) . row / ||row|| (float64 accumulate; the MFMA's fp32
    accumulation differs by < 1e-6)."""
    q16 = qn.astype(np.float16).astype(np.float64)
    r = rows16.astype(np.float64)
    nrm = np.sqrt((rows16.astype(np.float32).astype(np.float64) ** 2).sum(axis=1))
    return (r @ q16) / nrm


def build_case(oracle, k: int = 100, kprime: int = 128, rho: float = 0.2, n_background: int = 20_000, seed: int = 1):
    """Returns dict(corpus fp16 [n, 768], query fp32 [1, 768], victim row id, diagnostics).  `oracle` is
    oracle.canonical (test infrastructure)."""
    d = 768
    rng = np.random.default_rng(seed)
    q, _ = halfway_query(d)
    qn = oracle.normalize(q[None, :])[0]                          # canonical normalised query (fp32)
    delta = qn.astype(np.float16).astype(np.float64) - qn.astype(np.float64)
    qd = qn.astype(np.float64)
    qd /= np.linalg.norm(qd)
    dh = delta - (delta @ qd) * qd
    dh /= np.linalg.norm(dh)

    def make(rho_t, alpha, count):
        """rows with cosine rho_t to q and a component alpha * sqrt(1 - rho_t^2) along delta_hat"""
        z = rng.standard_normal((count, d))
        z -= np.outer(z @ qd, qd)
        z -= np.outer(z @ dh, dh)
        z /= np.linalg.norm(z, axis=1, keepdims=True)
        s = np.sqrt(1.0 - rho_t ** 2)
        rho_t = np.broadcast_to(np.asarray(rho_t, dtype=np.float64), (count,))
        s = np.broadcast_to(s, (count,))
        v = rho_t[:, None] * qd + s[:, None] * (alpha * dh + np.sqrt(1.0 - alpha ** 2) * z)
        return v.astype(np.float16)

    e_v = np.linalg.norm(delta) * np.sqrt(1.0 - rho ** 2)         # the victim's understatement
    victim = make(rho, -1.0, 1)
    can_v = float(oracle.cosine(q[None, :], victim.astype(np.float32))[0, 0])
    app_v = float(approx_scores(qn, victim)[0])
    eps_old, eps_new = eps_bound(d, 2.0 ** -12), eps_bound(d, 2.0 ** -11)
    # candidates for the strong rows (canonical in (app_v + eps_old + slack, can_v)) and the fillers (approx just above
    # app_v), generated with a spread of targets and SELECTED on their stored (fp16) values
    lo_s, hi_s = app_v + eps_old + 1.6e-5, can_v - 4e-6
    assert hi_s - lo_s > 1e-5, (lo_s, hi_s, e_v)
    cand = make(rng.uniform(lo_s - 1e-5, hi_s + 1e-5, 3000), 0.0, 3000)
    can_c = oracle.cosine(q[None, :], cand.astype(np.float32))[0]
    app_c = approx_scores(qn, cand)
    ok = (can_c > lo_s) & (can_c < hi_s) & (np.abs(app_c - can_c) < 2e-5)
    strong = cand[ok][:k]
    assert strong.shape[0] == k, f"only {int(ok.sum())} strong rows met the window"
    fill = make(rng.uniform(app_v + 2e-6, app_v + 2.4e-5, 1500), 0.0, 1500)
    app_f = approx_scores(qn, fill)
    okf = (app_f > app_v + 4e-6) & (app_f < app_v + 1.2e-5)
    filler = fill[okf][: (kprime - k) + 12]
    assert filler.shape[0] >= kprime - k + 4, f"only {int(okf.sum())} fillers met the window"
    background = rng.standard_normal((n_background, d)).astype(np.float16)
    rows = np.concatenate([background, filler, strong, victim])
    perm = rng.permutation(rows.shape[0])
    rows = np.ascontiguousarray(rows[perm])
    victim_id = int(np.nonzero(perm == rows.shape[0] - 1)[0][0])
    # what the certificate sees, emulated
    app = approx_scores(qn, rows)
    can = oracle.cosine(q[None, :], rows.astype(np.float32))[0]
    by_app = np.argsort(-app, kind="stable")[:kprime]
    A = float(app[by_app[-1]])
    ck_k = float(np.sort(can[by_app])[::-1][k - 1])
    return {"corpus": rows, "query": q[None, :].copy(), "victim": victim_id, "victim_rescored": bool(victim_id in by_app),
            "victim_canonical": can_v, "victim_approx": app_v, "A": A, "ck_k": ck_k, "eps_old": eps_old, "eps_new": eps_new,
            "true_best": int(np.argmax(can)), "understatement": e_v}
