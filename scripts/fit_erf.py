#!/usr/bin/env python3
"""Derives the coefficients of the branch-free erf used by the GEMM epilogue (vrd_common.h: erf_f32 / gelu_erf2).

    erf(z) = sign(z) * (1 - 2^(-G(u))),  u = min(|z|, 4),  G = -log2(erfc) fitted on [0, 4] (degree 11, least squares
    on Chebyshev nodes); erf rounds to 1 in f32 beyond 4.
One path for the whole range: near 0 the form loses RELATIVE accuracy (1 - 2^-G cancels) but keeps the absolute error at
one f32 rounding of 1, which is what GELU(x) = x/2 * (1 + erf(x / sqrt 2)) needs.
Prints the coefficients and the maximum error of an f32 Horner evaluation against scipy's f64 erf."""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P
from scipy.special import erf, erfc

DEG = 11


def cheb_fit(f, a, b, deg, n=600):
    k = np.arange(n)
    x = np.cos(np.pi * (k + 0.5) / n)
    t = 0.5 * (b - a) * x + 0.5 * (b + a)
    p = C.cheb2poly(C.chebfit(x, f(t), deg))
    lin = np.array([-(a + b) / (b - a), 2.0 / (b - a)])          # x as a polynomial in t
    out, powr = np.zeros(1), np.ones(1)
    for ck in p:
        out = P.polyadd(out, ck * powr)
        powr = P.polymul(powr, lin)
    return out


def horner32(c, x):
    acc = np.full_like(x, np.float32(c[-1]), dtype=np.float32)
    for ck in c[-2::-1]:
        acc = np.float32(acc * x + np.float32(ck))       # fma emulated in f32 (double rounding is negligible here)
    return acc


g = cheb_fit(lambda t: -np.log2(erfc(t)), 0.0, 4.0, DEG)
print("G (in |z|, constant term first):")
print(", ".join(f"{c:.9e}f" for c in g))

z = np.linspace(-6, 6, 2_000_001).astype(np.float32)
u = np.minimum(np.abs(z), np.float32(4.0))
e = np.sign(z) * (np.float32(1.0) - np.exp2(-horner32(g.astype(np.float32), u).astype(np.float64)).astype(np.float32))
ref = erf(z.astype(np.float64))
err = np.abs(e.astype(np.float64) - ref)
print("max abs erf err:", err.max(), "at z =", z[err.argmax()])
x = z.astype(np.float64) * np.sqrt(2.0)
gelu_ref, gelu = 0.5 * x * (1 + ref), 0.5 * x * (1 + e.astype(np.float64))
print("max abs gelu err:", np.abs(gelu - gelu_ref).max())
