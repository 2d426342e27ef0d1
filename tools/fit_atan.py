#!/usr/bin/env python
"""Coefficients of the float64 arctangent kernel used for edge lengths.

atan(t) = t + t*s*Q(s), s = t*t, t in [0, 1].  Q is interpolated at Chebyshev nodes of [0, 1]
in 60-digit arithmetic and converted to the monomial basis; the script prints the coefficients
as C hex floats and the maximum relative error of the float64 evaluation.
"""
import sys

import mpmath as mp
import numpy as np

mp.mp.dps = 60
N = int(sys.argv[1]) if len(sys.argv) > 1 else 15


def q(s):
    if s == 0:
        return -mp.mpf(1) / 3
    t = mp.sqrt(s)
    return (mp.atan(t) / t - 1) / s


nodes = [(mp.cos(mp.pi * (2 * k + 1) / (2 * N)) + 1) / 2 for k in range(N)]
vals = [q(x) for x in nodes]
# Newton divided differences -> monomial coefficients (exact enough at 60 digits)
coef = [mp.mpf(0)] * N
dd = vals[:]
for j in range(1, N):
    for i in range(N - 1, j - 1, -1):
        dd[i] = (dd[i] - dd[i - 1]) / (nodes[i] - nodes[i - j])
poly = [mp.mpf(0)] * N
basis = [mp.mpf(1)] + [mp.mpf(0)] * (N - 1)
for j in range(N):
    for i in range(N):
        poly[i] += dd[j] * basis[i]
    new = [mp.mpf(0)] * N
    for i in range(N - 1):
        new[i + 1] += basis[i]
    for i in range(N):
        new[i] -= nodes[j] * basis[i]
    basis = new
c = np.array([float(x) for x in poly])
t = np.concatenate([np.linspace(0, 1, 2_000_001), np.logspace(-12, 0, 100001)])
s = t * t
p = np.zeros_like(t)
for k in range(N - 1, -1, -1):
    p = p * s + c[k]
approx = t + t * s * p
exact = np.array([float(mp.atan(mp.mpf(float(x)))) for x in t[::50]])
rel = np.abs(approx[::50] - exact) / np.maximum(exact, 1e-300)
rel[t[::50] == 0] = 0
print("terms", N, "max rel err", rel.max())
for k, x in enumerate(c):
    print(f"  {float(x).hex()},  // s^{k}")
