"""TEST INFRASTRUCTURE: evaluate a BoundModel (the flat columns / terms / normaliser description the
host hands to gwi_create) in NumPy.  Lets the CPU suite validate the host-side logic -- column
transforms, static masks, normaliser grids, theta layout -- against the oracle and the golden
vectors without a GPU.  Value only; never imported by the product."""
import numpy as np
from scipy.special import betaln, erf

from gwinferno_amd import _native as N


def _taps(t):
    return np.stack([(1 - t) ** 3 / 6, (3 * t**3 - 6 * t**2 + 4) / 6, (-3 * t**3 + 3 * t**2 + 3 * t + 1) / 6, t**3 / 6])


def _spline(x, coefs, lo, hi, zero_outside):
    n = len(coefs)
    n_int = n - 3
    u = (x - lo) * (n_int / (hi - lo))
    k = np.clip(np.floor(u), 0, n_int - 1).astype(np.int64)
    b = _taps(u - k)
    v = sum(coefs[k + j] * b[j] for j in range(4))
    if zero_outside:
        v = np.where((x >= lo) & (x <= hi), v, 0.0)
    return v


def _pl_lognorm(alpha, lo, hi):
    a1 = 1.0 + alpha
    if a1 == 0:
        return -np.log(np.log(hi / lo))
    return np.log(a1 / (hi**a1 - lo**a1))


def _tn_lognorm(mu, sg, lo, hi):
    r2 = np.sqrt(2.0)
    d = 0.5 * (1 + erf((hi - mu) / (sg * r2))) - 0.5 * (1 + erf((lo - mu) / (sg * r2)))
    return -np.log(sg) - 0.5 * np.log(2 * np.pi) - np.log(d)


def norm_values(bm, theta):
    out = []
    for g, expo_theta, coef_off in bm.norms:
        e = np.zeros(len(g.tw)) if g.lb is None else g.lb.copy()
        if expo_theta >= 0:
            e = e + (theta[expo_theta] + g.expo_add) * g.l1
        if g.n_basis > 0 and (g.spline_flags & N.NORM_LINEAR_SPLINE):
            out.append(np.sum(g.tw * _spline(g.us, theta[coef_off : coef_off + g.n_basis], g.lo, g.hi, True)))
            continue
        if g.n_basis > 0:
            e = e + _spline(g.us, theta[coef_off : coef_off + g.n_basis], g.lo, g.hi, bool(g.spline_flags & N.SPLINE_OUTSIDE_ZERO_EXPONENT))
        with np.errstate(all="ignore"):
            out.append(np.sum(np.where(g.tw != 0, g.tw * np.exp(e), 0.0)))
    return np.array(out)


def log_weights(bm, theta, include_consts=True):
    """(pe_logw, inj_logw, norms): log importance weights incl. every normaliser.  With
    ``include_consts=False`` the sample-independent log-normalisers (power-law / truncated-normal /
    Beta constants and the grid normalisers) are left out, which is what the device scan sums."""
    theta = np.asarray(theta, dtype=np.float64)
    norms = norm_values(bm, theta)
    k_const = 1.0 if include_consts else 0.0
    outs = []
    for cols in (bm.pe_cols, bm.inj_cols):
        with np.errstate(all="ignore"):
            ell = cols[bm.kappa_col].copy()
            for t in bm.terms:
                c = [cols[i] for i in t["cols"]]
                th = [theta[i] for i in t["theta"]]
                p = t["p"]
                k = t["kind"]
                if k == N.TERM_POWERLAW and (t["flags"] & N.POWERLAW_UNNORMALISED):
                    ell = ell + th[0] * c[0]
                elif k == N.TERM_LINEAR_SPLINE:
                    co = t["coef_off"]
                    f = _spline(c[0], theta[co : co + t["n_basis"]], p[0], p[1], True)
                    ell = ell + np.log(np.where(f > 0, f, 0.0))
                elif k == N.TERM_TILT_JOINT:
                    xi, sg = th
                    ln = _tn_lognorm(1.0, sg, -1.0, 1.0)
                    A = np.exp(-0.5 * ((c[0] - 1) ** 2 + (c[1] - 1) ** 2) / sg**2 + 2 * ln)
                    ell = ell + np.log(0.25 * (1 - xi) + xi * A)
                elif k == N.TERM_POWERLAW:
                    ell = ell + th[0] * c[0] + k_const * _pl_lognorm(th[0], p[0], p[1])
                elif k == N.TERM_PLPEAK:
                    alpha, mu, sg, lam = th
                    pl = np.exp(alpha * c[0] + _pl_lognorm(alpha, p[0], p[1]))  # one column: log x
                    tn = np.exp(-0.5 * (np.exp(c[0]) - mu) ** 2 / sg**2 + _tn_lognorm(mu, sg, p[0], p[1]))
                    ell = ell + np.log((1 - lam) * pl + lam * tn)
                elif k == N.TERM_PLPEAK_SMOOTH:
                    alpha, mu, sg, lam, dl = th
                    y = c[0] - p[0]
                    taper = 1.0 / (1.0 + np.exp(dl / y + dl / (y - dl)))
                    pl = np.exp(alpha * c[1] + _pl_lognorm(alpha, p[0], p[1])) * taper
                    tn = np.exp(-0.5 * (c[0] - mu) ** 2 / sg**2 + _tn_lognorm(mu, sg, p[0], p[1]))
                    ell = ell + np.log((1 - lam) * pl + lam * tn)
                elif k == N.TERM_SMOOTH:
                    ell = ell - np.log1p(np.exp(th[0] / c[0] + th[0] / (c[0] - th[0])))
                elif k == N.TERM_POWERLAW_RATIO:
                    lr = p[0] - c[1]
                    b1 = 1 + th[0]
                    if b1 == 0:
                        ell = ell - c[0] - np.log(-lr)
                    else:
                        ell = ell + th[0] * c[0] + np.log(b1 / (-np.expm1(b1 * lr)))
                elif k == N.TERM_BETA:
                    ell = ell + (th[0] - 1) * c[0] + (th[1] - 1) * c[1] - k_const * betaln(th[0], th[1])
                elif k == N.TERM_TILT_MIXTURE:
                    xi, sg = th
                    tn = np.exp(-0.5 * (c[0] - 1) ** 2 / sg**2 + _tn_lognorm(1.0, sg, -1.0, 1.0))
                    ell = ell + np.log(0.5 * (1 - xi) + xi * tn)
                elif k == N.TERM_TRUNCNORM:
                    ell = ell - 0.5 * (c[0] - th[0]) ** 2 / th[1] ** 2 + k_const * _tn_lognorm(th[0], th[1], p[0], p[1])
                elif k == N.TERM_POWERLAW_REDSHIFT:
                    ell = ell + (th[0] - 1) * c[0]
                elif k == N.TERM_EXP_SPLINE:
                    co = t["coef_off"]
                    ell = ell + _spline(c[0], theta[co : co + t["n_basis"]], p[0], p[1], bool(t["flags"] & N.SPLINE_OUTSIDE_ZERO_EXPONENT))
                elif k == N.TERM_POWERLAW_BOUNDS:
                    alpha, lo, hi = th
                    inside = ~((c[1] < lo) | (c[1] > hi))
                    ln = -np.log(hi / lo) if alpha == -1.0 else _pl_lognorm(alpha, lo, hi)  # numpyro_distributions.py:130 as written
                    ell = ell + np.where(inside, alpha * c[0] + k_const * ln, -np.inf)
                elif k == N.TERM_EXP_SPLINE_LERP:
                    co = t["coef_off"]
                    g = bm.norms[t["norm"]][0]
                    zero_out = bool(t["flags"] & N.SPLINE_OUTSIDE_ZERO_EXPONENT)
                    lp = _spline(g.us, theta[co : co + t["n_basis"]], p[0], p[1], zero_out)
                    if not zero_out:
                        lp = np.where((g.us >= p[0]) & (g.us <= p[1]), lp, -np.inf)
                    j = np.clip(np.floor(c[0]).astype(np.int64), 0, len(lp) - 2)
                    f = c[0] - j
                    ell = ell + np.where(f == 0, lp[j], np.where(f == 1, lp[j + 1], (1 - f) * lp[j] + f * lp[j + 1]))
                else:
                    raise ValueError(k)
                if t["norm"] >= 0 and include_consts:
                    ell = ell - np.log(norms[t["norm"]])
            ell = np.where(ell < np.inf, ell, -np.inf)  # NaN / +inf -> excluded
        outs.append(ell)
    return outs[0], outs[1], norms
