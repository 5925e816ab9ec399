#!/usr/bin/env python3
"""Derives the coefficients of the branch-free erf used by the GEMM epilogue (vrd_common.h: erf_f32).

    |z| <= 1 : erf(z) = z * P(z^2)                         (P: degree 7, least squares on Chebyshev nodes)
    |z| >  1 : erf(z) = sign(z) * (1 - 2^(-G(|z|)))        (G = -log2(erfc): degree 9 on [1, 4]; |z| clamped to 4)
Prints the coefficients and the maximum error of an f32 Horner evaluation against scipy's f64 erf."""
import numpy as np
from numpy.polynomial import chebyshev as C, polynomial as P
from scipy.special import erf, erfc

def cheb_fit(f, a, b, deg, n=400):
    k = np.arange(n)
    x = np.cos(np.pi * (k + 0.5) / n)
    t = 0.5 * (b - a) * x + 0.5 * (b + a)
    c = C.chebfit(x, f(t), deg)
    # to monomial in t
    p = C.cheb2poly(c)
    # substitute x = (2t - (a+b)) / (b-a)
    lin = np.array([-(a + b) / (b - a), 2.0 / (b - a)])
    out = np.zeros(1)
    powr = np.ones(1)
    for ck in p:
        out = P.polyadd(out, ck * powr)
        powr = P.polymul(powr, lin)
    return out

def horner32(c, x):
    acc = np.full_like(x, np.float32(c[-1]), dtype=np.float32)
    for ck in c[-2::-1]:
        acc = np.float32(acc * x + np.float32(ck))       # fma emulated in f32 (double rounding is negligible here)
    return acc

pa = cheb_fit(lambda t: erf(np.sqrt(t)) / np.sqrt(t), 1e-12, 1.0, 7)
pb = cheb_fit(lambda u: -np.log2(erfc(u)), 1.0, 4.0, 9)
print("A (in z^2):", ", ".join(f"{c:.9e}f" for c in pa))
print("B (in |z|):", ", ".join(f"{c:.9e}f" for c in pb))

z = np.linspace(-6, 6, 2_000_001).astype(np.float32)
az = np.abs(z)
ea = z * horner32(pa.astype(np.float32), (z * z).astype(np.float32))
u = np.minimum(az, np.float32(4.0))
g = horner32(pb.astype(np.float32), u)
eb = np.sign(z) * (np.float32(1.0) - np.exp2(-g.astype(np.float64)).astype(np.float32))
e = np.where(az <= 1.0, ea, eb).astype(np.float64)
ref = erf(z.astype(np.float64))
err = np.abs(e - ref)
print("max abs err:", err.max(), "at z =", z[err.argmax()])
x = z.astype(np.float64) * np.sqrt(2.0)
gelu_ref = 0.5 * x * (1 + ref)
gelu = 0.5 * x * (1 + e)
print("max abs gelu err:", np.abs(gelu - gelu_ref).max(), " max rel (|x|<4):", (np.abs(gelu - gelu_ref) / np.maximum(np.abs(gelu_ref), 1e-30))[np.abs(x) < 4].max())
