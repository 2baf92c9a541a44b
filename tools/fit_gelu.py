#!/usr/bin/env python3
"""Fit and exhaustive check of the GELU the HIP epilogues use (vf_transformer.hip: gelu_erf):
    gelu(x) = max(x, 0) - |x| * exp2(-(a * h(a)) - 1),  a = |x|,  h = c1 + a (c2 + a (c3 + a (c4 + a c5)))
i.e. -log2 erfc(a / sqrt 2) as a degree-5 polynomial without constant term, weighted least squares + Lawson iterations
on [0, 6] with the weight (a / 2) erfc(a / sqrt 2) ln 2 of the term the exponential multiplies.  Prints the coefficients and
the error of an fp32 evaluation over EVERY finite fp16 input."""
import numpy as np
from scipy.special import erf, erfc

A = 6.0
n = 8000
a = (np.cos(np.pi * (np.arange(n) + 0.5) / n) * 0.5 + 0.5) * A
p = -np.log2(erfc(a / np.sqrt(2)))
w = (a / 2) * erfc(a / np.sqrt(2)) * np.log(2) + 1e-7
V = np.stack([a ** k for k in range(1, 6)], axis=1)
lw = w.copy()
c = np.linalg.lstsq(V * lw[:, None], p * lw, rcond=None)[0]
for _ in range(30):
    r = np.abs((V @ c - p) * w)
    lw = lw * (r / r.mean() + 1e-3) ** 0.5
    c = np.linalg.lstsq(V * lw[:, None], p * lw, rcond=None)[0]
c = c.astype(np.float32)
print("c1..c5 =", ", ".join("%.9ef" % v for v in c))
h16 = np.frombuffer(np.arange(0, 0x7c00, dtype=np.uint16).tobytes(), dtype=np.float16).astype(np.float32)
x = np.concatenate([h16, -h16]).astype(np.float32)
ax = np.abs(x)
h = np.full_like(ax, c[4])
for ck in c[3::-1]:
    h = (h * ax + ck).astype(np.float32)
s = (-(ax * h) - np.float32(1.0)).astype(np.float32)
e = np.exp2(s.astype(np.float64)).astype(np.float32)
g = (np.maximum(x, 0) - ax * e).astype(np.float32)
ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
err = np.abs(g - ref)
print("max |error| over all %d fp16 inputs: %.3e at x = %g" % (len(x), err.max(), x[err.argmax()]))
