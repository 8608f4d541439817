#!/usr/bin/env python
"""Coefficients and error scan of `gelu_erf_fast` (afford-motion_amd/csrc/common.h): GELU(x) = max(x, 0) - |x| 2^P(min(|x|, 5.5)), P = the degree-9
least-squares fit of log2(erfc(a / sqrt 2) / 2) at 6000 Chebyshev nodes of [0, 5.5].  Prints the float32 coefficients (low order first) and the
maximum absolute error against float64 over 3 M float32 inputs, evaluated with float32 FMAs as the kernel does, next to round 3's
Abramowitz & Stegun 7.1.26 form.  CPU only (numpy + scipy):  python tools/fit_gelu.py"""
import numpy as np
from scipy.special import erf, erfc

A, DEG = 5.5, 9
k = np.arange(6000)
a = (np.cos(np.pi * (k + 0.5) / 6000) + 1) / 2 * A
p = np.polynomial.chebyshev.Chebyshev.fit(a, np.log2(erfc(a / np.sqrt(2)) / 2), DEG, domain=[0, A]).convert(kind=np.polynomial.Polynomial).coef.astype(np.float32)
print("coefficients, low order first:", ", ".join(f"{v:.9e}f" for v in p))
xs = np.concatenate([np.linspace(-9, 9, 2000001), np.random.default_rng(0).normal(size=1000000) * 2]).astype(np.float32)
exact = 0.5 * xs.astype(np.float64) * (1 + erf(xs.astype(np.float64) / np.sqrt(2)))
fma = lambda x, y, z: np.float32(np.float64(x) * np.float64(y) + np.float64(z))          # one rounding
ax = np.minimum(np.abs(xs), np.float32(A))
acc = np.full_like(ax, p[-1])
for c in p[-2::-1]:
    acc = fma(acc, ax, np.float32(c))
new = fma(-np.abs(xs), np.exp2(acc.astype(np.float64)).astype(np.float32), np.maximum(xs, 0))
t = np.float32(1 / fma(np.float32(0.3275911 * 0.70710678118654752440), np.abs(xs), np.float32(1)))
pl = fma(np.float32(0.5 * 1.061405429), t, np.float32(0.5 * -1.453152027))
for c in (0.5 * 1.421413741, 0.5 * -0.284496736, 0.5 * 0.254829592):
    pl = fma(pl, t, np.float32(c))
old = fma(-np.abs(xs), np.float32(np.float64(pl) * t * np.exp2(np.float64(np.float32(xs * xs * np.float32(-0.72134752044448170368))))), np.maximum(xs, 0))
for name, g in (("2^P(|x|) form (round 4)", new), ("A&S 7.1.26 form (round 3)", old)):
    e = np.abs(g - exact)
    print(f"{name}: max |error| {e.max():.3e} at x = {xs[e.argmax()]:.4f}")
