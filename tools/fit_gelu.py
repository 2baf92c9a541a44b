#!/usr/bin/env python3
"""Fit and exhaustive check of the GELUs the HIP epilogues use (vf_transformer.hip: gelu_erf; with --tanh: gelu_tanh, the
same form for gemma's gelu_pytorch_tanh, -log2 sigmoid(-2 u(a)) - 1 = a h(a) fitted on [0, 5.5]):
    gelu(x) = max(x, 0) - |x| * exp2(-(a * h(a)) - 1),  a = |x|,  h = c1 + a (c2 + a (c3 + a (c4 + a c5)))
i.e. -log2 erfc(a / sqrt 2) as a degree-5 polynomial without constant term, weighted least squares + Lawson iterations
on [0, 6] with the weight (a / 2) erfc(a / sqrt 2) ln 2 of the term the exponential multiplies.  Prints the coefficients and
the error of an fp32 evaluation over EVERY finite fp16 input."""
import sys
import numpy as np
from scipy.special import erf, erfc

TANH = "--tanh" in sys.argv
K0, K1 = 0.7978845608028654, 0.044715
def u2(a): return 2 * K0 * (a + K1 * a ** 3)
A = 5.5 if TANH else 6.0
n = 8000
a = (np.cos(np.pi * (np.arange(n) + 0.5) / n) * 0.5 + 0.5) * A
if TANH:
    p = np.log2(1 + np.exp(u2(a))) - 1.0
    w = a / (1 + np.exp(u2(a))) * np.log(2) + 1e-8
else:
    p = -np.log2(erfc(a / np.sqrt(2)))
    w = (a / 2) * erfc(a / np.sqrt(2)) * np.log(2) + 1e-7
V = np.stack([a ** k for k in range(1, 6)], axis=1)
lw = w.copy()
c = np.linalg.lstsq(V * lw[:, None], p * lw, rcond=None)[0]
for _ in range(40 if TANH else 30):
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
e = np.exp2(np.maximum(s.astype(np.float64), -1000.0)).astype(np.float32)
g = (np.maximum(x, 0) - ax * e).astype(np.float32)
xd = x.astype(np.float64)
if TANH:
    with np.errstate(over="ignore"):
        ref = xd / (1 + np.exp(-np.sign(xd) * u2(np.abs(xd))))
else:
    ref = 0.5 * xd * (1 + erf(xd / np.sqrt(2)))
err = np.abs(g - ref)
print("max |error| over all %d fp16 inputs: %.3e at x = %g" % (len(x), err.max(), x[err.argmax()]))
