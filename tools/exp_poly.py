#!/usr/bin/env python3
"""Coefficients of the scan kernels' exponential (gwinferno_amd/csrc/gwi_device.h: GWI_EXP_C0..C8).

e^x = 2^n e^r, n = rint(x log2 e), r = x - n ln2, |r| <= ln2 / 2; e^r = 1 + r q(r) with q a polynomial of degree d for
(e^r - 1) / r.  The coefficients minimise the largest RELATIVE error of e^r over |r| <= 0.3466 (Lawson's reweighted least
squares on a fine grid, long double arithmetic); the script prints them with the achieved error per degree, and the error of
the one-piece reduction r = fma(n, -ln2_double, x).      python tools/exp_poly.py [degree=8]"""
import sys

import numpy as np

LD = np.longdouble
L = LD(0.3466)


def q_exact(r):
    r = np.asarray(r, dtype=LD)
    out = np.empty_like(r)
    small = np.abs(r) < 1e-4
    rs = r[small]
    out[small] = 1 + rs / 2 + rs**2 / 6 + rs**3 / 24 + rs**4 / 120 + rs**5 / 720
    out[~small] = np.expm1(r[~small]) / r[~small]
    return out


def minimax(degree, n_grid=20001, iterations=80):
    r = np.linspace(-1, 1, n_grid).astype(LD) * L
    A = np.polynomial.polynomial.polyvander(r, degree).astype(LD)
    weight = np.abs(r) * np.exp(-r)  # relative error of e^r = |r (q - q*)| e^-r
    lawson = np.ones_like(r)
    target = q_exact(r)
    best = None
    for _ in range(iterations):
        W = lawson * weight
        c = np.linalg.lstsq((A * W[:, None]).astype(np.float64), (target * W).astype(np.float64), rcond=None)[0].astype(LD)
        for _ in range(2):  # refine the float64 solve in long double
            res = (target - A @ c) * W
            c = c + np.linalg.lstsq((A * W[:, None]).astype(np.float64), res.astype(np.float64), rcond=None)[0].astype(LD)
        err = np.abs((A @ c - target) * weight)
        if best is None or err.max() < best[0]:
            best = (err.max(), c.copy())
        lawson = lawson * (1 + 3 * err / err.max())
        lawson /= lawson.max()
    return best


def fp64_error(c64, n_grid=200001):
    """largest relative error of 1 + r q(r) with q in fp64 Horner form against long double e^r"""
    r = (np.linspace(-1, 1, n_grid) * float(L)).astype(np.float64)
    q = np.full_like(r, c64[-1])
    for cj in c64[-2::-1]:
        q = q * r + cj
    approx = (1.0 + r * q).astype(LD)
    exact = np.exp(r.astype(LD))
    return float(np.max(np.abs((approx - exact) / exact)))


if __name__ == "__main__":
    degrees = [int(a) for a in sys.argv[1:]] or [8]
    for d in degrees:
        err, c = minimax(d)
        c64 = np.asarray(c, dtype=np.float64)
        print(f"degree {d}: minimax relative error of e^r {float(err):.3e}; evaluated in fp64 Horner form {fp64_error(c64):.3e}")
        for j, cj in enumerate(c64):
            print(f"#define GWI_EXP_C{j} {cj:.17e}")
    ln2_d = np.float64(0.6931471805599453)
    ln2 = LD("0.693147180559945309417232121458176568")
    print(f"ln2 as a double is off by {float(LD(ln2_d) - ln2):.3e}: the one-piece reduction moves r by that times n, i.e. e^x by {abs(float(LD(ln2_d) - ln2)) * 1010:.1e} relative at |x| = 700")
    taylor = np.array([1.0 / np.prod(np.arange(1, k + 2, dtype=np.float64)) for k in range(11)])
    print(f"for comparison, the Taylor series of degree 10 the kernels used through round 5: {fp64_error(taylor):.3e}")
