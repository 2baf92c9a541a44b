"""The exactness certificate's error bound, attacked on the CPU: no GPU needed.

`adversarial.build_case` makes a corpus on which the round-1 bound (fp16 unit round-off taken as 2^-12) would have
certified a WRONG top-k -- the true best match is never re-scored -- while the rigorous 2^-11 bound refuses to certify
(the GPU test then checks that the library repairs the query: tests/test_gpu_retrieval.py).
"""
import numpy as np

import adversarial as ADV


def test_halfway_query_rounding_error_exceeds_2_pow_minus_12(oracle):
    q, sign = ADV.halfway_query()
    qn = oracle.normalize(q[None, :])[0]
    assert abs(float(np.linalg.norm(qn.astype(np.float64))) - 1.0) < 2e-6
    delta = qn.astype(np.float16).astype(np.float64) - qn.astype(np.float64)
    # every element moved the way it was built to (down for sign -1, up for +1), by ~2^-11 / mantissa of its magnitude
    assert np.all(np.sign(delta * np.sign(qn)) == sign)
    rel = np.abs(delta) / np.abs(qn)
    assert rel.min() > 3.8e-4 and rel.max() < 2.0 ** -11
    # a row parallel to fp16(qn) is mis-scored by more than the old bound allowed when all elements round ONE way
    # (shrinking the "below" half a little renormalises every element to just above its half-way point: all round up)
    q1 = np.abs(q) * (1.0 - 6e-5 * (sign > 0))
    qn1 = oracle.normalize(q1[None, :])[0]
    row = qn1.astype(np.float16)
    can = float(oracle.cosine(q1[None, :], row.astype(np.float32)[None, :])[0, 0])
    app = float(ADV.approx_scores(qn1, row[None, :])[0])
    assert ADV.eps_bound(768, 2.0 ** -12) < abs(can - app) < ADV.eps_bound(768, 2.0 ** -11)


def test_old_bound_certifies_a_wrong_topk_and_new_bound_does_not(oracle):
    c = ADV.build_case(oracle)
    assert c["true_best"] == c["victim"] and not c["victim_rescored"]
    assert c["victim_canonical"] - c["victim_approx"] > c["eps_old"]          # the bound that was not a bound
    assert c["victim_canonical"] - c["victim_approx"] < c["eps_new"]
    assert c["ck_k"] > c["A"] + c["eps_old"], "construction lost its teeth: the old certificate no longer passes"
    assert not (c["ck_k"] > c["A"] + c["eps_new"]), "the rigorous bound must refuse to certify this query"
    ids, _ = oracle.search(c["corpus"], c["query"], 100)
    assert ids[0, 0] == c["victim"]
