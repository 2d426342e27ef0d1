#!/usr/bin/env python
"""Coefficients of a float32 arctangent kernel for edge lengths, and the error of the whole float32 pipeline
2 * atan2(sqrt(a), sqrt(1 - a)) against the exact value, over the range of the haversine term a.

atan(t) = t + t*s*P(s), s = t*t, t = min/max in [0, 1]; P interpolated at Chebyshev nodes of [0, 1] in 60-digit
arithmetic.  float32 fused multiply-adds are emulated through float64 (exact product and sum, one rounding).

    python tools/fit_atan_f32.py [terms]
"""
import sys

import mpmath as mp
import numpy as np

mp.mp.dps = 60
N = int(sys.argv[1]) if len(sys.argv) > 1 else 9
f32 = np.float32


def q(s):
    if s == 0:
        return -mp.mpf(1) / 3
    t = mp.sqrt(s)
    return (mp.atan(t) / t - 1) / s


nodes = [(mp.cos(mp.pi * (2 * k + 1) / (2 * N)) + 1) / 2 for k in range(N)]
vals = [q(x) for x in nodes]
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
coef = np.array([float(x) for x in poly]).astype(f32)  # s^0 .. s^(N-1)


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def atan2_pos(y, x):
    hi, lo = np.maximum(x, y), np.minimum(x, y)
    r = (1.0 / hi.astype(np.float64)).astype(f32)  # v_rcp_f32: within one ulp; one Newton step below
    r = fma(fma(-hi, r, np.ones_like(r)), r, r)
    t = (lo * r).astype(f32)
    t = fma(fma(-hi, t, lo), r, t)
    s = (t * t).astype(f32)
    p = np.full_like(t, coef[N - 1])
    for k in range(N - 2, -1, -1):
        p = fma(p, s, np.full_like(t, coef[k]))
    at = fma((t * s).astype(f32), p, t)
    pio2_hi, pio2_lo = f32(1.5707964), f32(-4.371139e-8)
    swapped = ((pio2_hi - at).astype(f32) + pio2_lo).astype(f32)
    return np.where(y > x, swapped, at)


rng = np.random.default_rng(0)
a = np.concatenate([10.0 ** rng.uniform(-16, 0, 3_000_000), rng.uniform(0, 1, 3_000_000), 1 - 10.0 ** rng.uniform(-8, 0, 500_000)])
a = np.clip(a, 0, 1).astype(f32)
y, x = np.sqrt(a), np.sqrt((f32(1) - a).astype(f32))
got = (f32(2) * atan2_pos(y, x)).astype(np.float64)
exact = 2.0 * np.arctan2(y.astype(np.float64), x.astype(np.float64))
ok = exact > 0
rel = np.abs(got[ok] - exact[ok]) / exact[ok]
ulp = np.abs(got[ok] - exact[ok]) / np.spacing(exact[ok].astype(f32)).astype(np.float64)
print(f"terms {N}: max relative error {rel.max():.3e} ({ulp.max():.2f} ulp), mean {ulp.mean():.3f} ulp, "
      f"correctly rounded {100.0 * np.mean(got[ok] == exact[ok].astype(f32).astype(np.float64)):.2f} %")
ref = (f32(2) * np.arctan2(y, x)).astype(np.float64)  # numpy's own float32 arctan2 on this machine
print(f"numpy float32 arctan2: max relative error {(np.abs(ref[ok] - exact[ok]) / exact[ok]).max():.3e}, "
      f"max difference to this kernel {(np.abs(ref[ok] - got[ok]) / exact[ok]).max():.3e}")
for k, c in enumerate(coef):
    print(f"  {float(c).hex()}f,  // s^{k}")
