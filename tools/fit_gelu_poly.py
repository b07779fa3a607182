"""Coefficients of KX_ACT_GELU_POLY (csrc/kx_common.h): 0.5*erf(x/sqrt2) ~ u*Q(u^2), u = clamp(x, +-C)/C, C = 3*sqrt(2).

Q is a degree-8 polynomial fitted in the Chebyshev basis by Lawson-weighted least squares (weights = the error it causes
on the GELU output x*(0.5 + u*Q)), converted to monomials in s = u^2 in [0, 1] and checked with fp32 Horner arithmetic.
Prints the fp32 coefficients (constant first) and the worst error on the GELU output over [-9, 9]."""
import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as P
from scipy.special import erf

DEG, C = 8, 3.0 * np.sqrt(2)


def fit(deg=DEG, c=C, iters=300):
    u = (np.cos(np.linspace(0, np.pi, 8001)) + 1) / 2
    u = u[u > 1e-5]
    z = 2 * u * u - 1
    y = 0.5 * erf(u * c / np.sqrt(2)) / u
    V = Ch.chebvander(z, deg)
    w = np.ones_like(u)
    for _ in range(iters):
        coef, *_ = np.linalg.lstsq(V * w[:, None], y * w, rcond=None)
        r = np.abs(V @ coef - y) * u * np.maximum(u * c, 0.3)
        w = w * (1 + 1.5 * r / r.max())
        w /= w.mean()
    pz = Ch.cheb2poly(coef)
    ps = np.zeros(1)
    for k, a in enumerate(pz):
        ps = P.polyadd(ps, a * P.polypow([-1, 2], k))
    return ps


def check(ps, c=C):
    xs = np.linspace(-9, 9, 400001).astype(np.float32)
    c32 = ps.astype(np.float32)
    u = (np.clip(xs, -np.float32(c), np.float32(c)) * np.float32(1 / c)).astype(np.float32)
    s = (u * u).astype(np.float32)
    q = np.full_like(xs, c32[-1])
    for k in range(len(c32) - 2, -1, -1):
        q = (q * s + c32[k]).astype(np.float32)
    g = (xs * (q * u + np.float32(0.5)).astype(np.float32)).astype(np.float32)
    gt = 0.5 * xs.astype(np.float64) * (1 + erf(xs.astype(np.float64) / np.sqrt(2)))
    return float(np.abs(g - gt).max())


if __name__ == "__main__":
    ps = fit()
    print("C =", float(np.float32(C)), " 1/C =", float(np.float32(1 / C)))
    print("Q (constant first):", [float(v) for v in ps.astype(np.float32)])
    print("max |gelu_poly - gelu| on [-9, 9], fp32 Horner: %.2e" % check(ps))
