"""Posterior-predictive density curves on the engine, with the reference's
function names, arguments, grids and return values (gwinferno/postprocess/calculations.py:20-276: all eight functions).

For every posterior draw the reference evaluates the population density on an 800 x 800 (m1, q) mesh and
integrates it along each axis with the trapezoid rule (:44-52, :78-84).  That is the likelihood hot path
on a synthetic catalog: take the mesh rows as "events" and the trapezoid weights as the inverse sampling
prior, and the engine's per-event importance sums ARE the marginals --

    p_q(q_i) = trapz_m p(m, q_i)  =  N_pe * exp(logBF_i)        (rows = q, samples = m)
    p_m(m_j) = trapz_q p(m_j, q)  =  the same on the transposed mesh

so K draws are K hyper-parameter points of ``gwi_eval_batch``: no new kernels, no (N_draws, 800, 800)
temporaries, and the B-spline models never build their (N_basis, 640 000) design matrices.  The 1-D curves
(spin magnitudes, tilts) are the engine's per-sample log-weights (``gwi_log_weights``) on an 800-point grid, and so are
the merger-rate curves R(z) (:244-276) on the redshift model's own 1000-point grid ``z_model.zs``.

``rate`` / ``pop_frac`` scale the normalised curves exactly as the reference does (:50-51).
"""
import numpy as np

from . import _native as N
from . import models as M
from .engine import NativePopulationLikelihood
from .interpolation import LogYBSpline, trapezoid_weights
from .lazy import Column, Density, Factor, side_of

GRID = 800  # points per axis in every reference PPD function


def _ones(like, n):
    return np.ones(n) if like is None else np.asarray(like, dtype=np.float64)


class _MeshMarginals:
    """Two engines over one (ys x xs) mesh: per-row sums over x, and (transposed) per-column sums over y."""

    def __init__(self, xs, ys, weights_fn, placeholder, keep=None):
        self.xs, self.ys = np.asarray(xs, dtype=np.float64), np.asarray(ys, dtype=np.float64)
        X, Y = np.meshgrid(self.xs, self.ys)  # X[i, j] = xs[j], Y[i, j] = ys[i]   (calculations.py:24)
        twx, twy = trapezoid_weights(self.xs), trapezoid_weights(self.ys)
        keep = np.ones_like(X) if keep is None else keep(X, Y).astype(np.float64)
        # orientation A: events = rows (y), samples = x -> integrates over x;  B: the transpose
        self.sides = []
        for Xa, Ya, tw in ((X, Y, twx[None, :] * keep), (np.ascontiguousarray(X.T), np.ascontiguousarray(Y.T), twy[None, :] * keep.T)):
            fn = weights_fn(Xa, Ya, self.xs, self.ys)  # -> callable(draw, pe_samples) -> lazy density
            pe_w = lambda d, fn=fn, tw=np.ascontiguousarray(tw): fn(d, True) * tw  # noqa: E731
            eng = NativePopulationLikelihood(pe_w(placeholder), fn(placeholder, False))
            self.sides.append((pe_w, eng))

    def __call__(self, draws):
        """draws: list of parameter dicts -> (over_x[n, len(ys)], over_y[n, len(xs)])"""
        out = []
        for pe_w, eng in self.sides:
            thetas = np.stack([eng.bound.theta_of(pe_w(d)) for d in draws])
            rows = []
            kmax = 16
            for k0 in range(0, len(draws), kmax):
                res = eng.evaluate_batch(thetas[k0 : k0 + kmax], float(eng.n_inj), min_neff_cut=False, want_grad=False)
                rows.extend(np.exp(r.log_bfs) * eng.n_pe for r in res)
            out.append(np.array(rows))
        return out[0], out[1]

    def close(self):
        for _, eng in self.sides:
            eng.close()


def _normalise(p, grid, rate, frac):
    return rate[:, None] * p * frac[:, None] / np.trapezoid(p, grid, axis=1)[:, None]


def _mass_ppds(weights_fn, draws, placeholder, mmin, mmax, rate, pop_frac, keep=None):
    ms = np.linspace(mmin, mmax, GRID)
    qs = np.linspace(mmin / mmax, 1, GRID)
    keep = keep if keep is not None else (lambda Mg, Qg: Qg > mmin / Mg)  # calculations.py:46, 80
    mesh = _MeshMarginals(ms, qs, weights_fn, placeholder, keep=keep)
    p_q, p_m = mesh(draws)  # integrate over m (axis=1 of the mesh) / over q (axis=0)
    mesh.close()
    n = len(draws)
    rate, pop_frac = _ones(rate, n), _ones(pop_frac, n)
    return _normalise(p_m, ms, rate, pop_frac), ms, _normalise(p_q, qs, rate, pop_frac), qs


def calculate_powerlaw_peak_mass_ppds(alpha, beta, mu_peak, sig_peak, lamb, mmin, mmax, rate=None, pop_frac=None):
    """calculations.py:63-91 -> ``(mpdfs, ms, qpdfs, qs)``."""
    cols = [np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in (alpha, beta, mu_peak, sig_peak, lamb)]
    draws = [dict(zip(("a", "b", "mp", "sp", "lam"), (float(c[i]) for c in cols))) for i in range(len(cols[0]))]

    def weights_fn(Mg, Qg, ms, qs):
        data = {True: (Mg, Qg), False: (ms, qs)}

        def w(d, pe_samples):
            m1, q = data[pe_samples]
            return M.plpeak_primary_ratio_pdf(m1, q, d["a"], d["b"], mmin, mmax, d["mp"], d["sp"], d["lam"])

        return w

    return _mass_ppds(weights_fn, draws, dict(a=-2.0, b=1.0, mp=30.0, sp=5.0, lam=0.1), mmin, mmax, rate, pop_frac)


def calculate_bspline_mass_ppds(m_cs, q_cs, nspline_dict, mmin, mmax, rate=None, pop_frac=None):
    """calculations.py:20-60 -> ``(mpdfs, ms, qpdfs, qs)``."""
    m_cs, q_cs = np.atleast_2d(np.asarray(m_cs, dtype=np.float64)), np.atleast_2d(np.asarray(q_cs, dtype=np.float64))
    draws = [dict(m=m_cs[i], q=q_cs[i]) for i in range(m_cs.shape[0])]

    def weights_fn(Mg, Qg, ms, qs):
        model = M.BSplinePrimaryBSplineRatio(nspline_dict["m1"], nspline_dict["q"], Mg, ms, Qg, qs, m1min=mmin, m2min=mmin, mmax=mmax)
        return lambda d, pe_samples: model(d["m"], d["q"], pe_samples=pe_samples)

    return _mass_ppds(weights_fn, draws, dict(m=np.zeros(nspline_dict["m1"]), q=np.zeros(nspline_dict["q"])), mmin, mmax, rate, pop_frac)


def calculate_peak_logm1_bspline_q_ppds(logmp, logsigp, q_cs, nspline_dict, mmin, mmax, rate=None, pop_frac=None):
    """calculations.py:94-130: log-normal primary-mass peak x LogY B-spline mass ratio -> ``(mpdfs, ms, qpdfs, qs)``."""
    logmp, logsigp = np.atleast_1d(np.asarray(logmp, dtype=np.float64)), np.atleast_1d(np.asarray(logsigp, dtype=np.float64))
    q_cs = np.atleast_2d(np.asarray(q_cs, dtype=np.float64))
    draws = [dict(mu=float(logmp[i]), sg=float(logsigp[i]), q=q_cs[i]) for i in range(q_cs.shape[0])]

    def weights_fn(Mg, Qg, ms, qs):
        q_model = M.BSplineRatio(nspline_dict["q"], Qg, qs, mmin / mmax, basis=LogYBSpline)
        data = {True: Mg, False: ms}
        return lambda d, pe_samples: q_model(d["q"], pe_samples=pe_samples) * M.truncnorm_pdf(data[pe_samples], d["mu"], d["sg"], mmin, mmax, log=True)

    keep = lambda Mg, Qg: ~((Mg < mmin) | (Mg * Qg < mmin))  # noqa: E731  (calculations.py:118)
    return _mass_ppds(weights_fn, draws, dict(mu=3.0, sg=0.5, q=np.zeros(nspline_dict["q"])), mmin, mmax, rate, pop_frac, keep=keep)


# ---- 1-D curves: per-sample log-weights of an 800-point grid -------------------------------------------
def _curves(grid, density_fn, draws, placeholder):
    """density_fn(x, draw) -> lazy density of x; returns pdf[n_draws, len(grid)] (unnormalised)."""
    grid = np.asarray(grid, dtype=np.float64)
    pe = np.ascontiguousarray(grid[None, :])
    eng = NativePopulationLikelihood(density_fn(pe, placeholder), density_fn(grid, placeholder))
    out = []
    for d in draws:
        _, logw = eng.log_weights(eng.bound.theta_of(density_fn(pe, d)))
        out.append(np.exp(logw))
    eng.close()
    return np.array(out)


def calculate_beta_spin_mag(alpha_a, beta_a, amax=1, rate=None, pop_frac=None):
    """calculations.py:133-154 -> ``(apdfs, aa)``."""
    aa = np.linspace(0, amax, GRID)
    alpha_a, beta_a = np.atleast_1d(np.asarray(alpha_a, dtype=np.float64)), np.atleast_1d(np.asarray(beta_a, dtype=np.float64))
    draws = list(zip(alpha_a, beta_a))
    p = _curves(aa, lambda x, d: M.betadist(x, d[0], d[1], scale=amax), draws, (2.0, 2.0))
    n = len(draws)
    return _normalise(p, aa, _ones(rate, n), _ones(pop_frac, n)), aa


def calculate_mixture_iso_aligned_spin_tilt(sig_ct, lambda_ct, rate=None, pop_frac=None):
    """calculations.py:157-178 -> ``(ctpdfs, ct)``."""
    ct = np.linspace(-1, 1, GRID)
    sig_ct, lambda_ct = np.atleast_1d(np.asarray(sig_ct, dtype=np.float64)), np.atleast_1d(np.asarray(lambda_ct, dtype=np.float64))
    draws = list(zip(sig_ct, lambda_ct))
    p = _curves(ct, lambda x, d: M.mixture_isoalign_spin_tilt(x, d[1], d[0]), draws, (1.0, 0.5))
    n = len(draws)
    return _normalise(p, ct, _ones(rate, n), _ones(pop_frac, n)), ct


def calculate_bspline_spin_ppds(a1_cs, tilt1_cs, nspline_dict, a2_cs=None, tilt2_cs=None, rate=None, pop_frac=None):
    """calculations.py:181-241: IID form -> ``(apdfs, aa, ctpdfs, cc)``; independent form ->
    ``(apdfs_1, apdfs_2, aa, ctpdfs_1, ctpdfs_2, cc)`` (LogYBSpline bases, normalised)."""
    aa, cc = np.linspace(0, 1, GRID), np.linspace(-1, 1, GRID)
    a1_cs, tilt1_cs = np.atleast_2d(np.asarray(a1_cs, dtype=np.float64)), np.atleast_2d(np.asarray(tilt1_cs, dtype=np.float64))
    n = a1_cs.shape[0]
    rate, pop_frac = _ones(rate, n), _ones(pop_frac, n)

    def spin_curves(grid, cls, n_splines, coefs):
        model = cls(n_splines, grid[None, :], grid, basis=LogYBSpline, normalize=True)
        dens = lambda x, c: model(c, pe_samples=np.ndim(x) == 2)  # noqa: E731
        return _normalise(_curves(grid, dens, list(coefs), np.zeros(n_splines)), grid, rate, pop_frac)

    if a2_cs is None:
        return (spin_curves(aa, M.BSplineSpinMagnitude, nspline_dict["a"], a1_cs), aa,
                spin_curves(cc, M.BSplineSpinTilt, nspline_dict["tilt"], tilt1_cs), cc)
    a2_cs, tilt2_cs = np.atleast_2d(np.asarray(a2_cs, dtype=np.float64)), np.atleast_2d(np.asarray(tilt2_cs, dtype=np.float64))
    return (spin_curves(aa, M.BSplineSpinMagnitude, nspline_dict["a1"], a1_cs), spin_curves(aa, M.BSplineSpinMagnitude, nspline_dict["a2"], a2_cs), aa,
            spin_curves(cc, M.BSplineSpinTilt, nspline_dict["tilt1"], tilt1_cs), spin_curves(cc, M.BSplineSpinTilt, nspline_dict["tilt2"], tilt2_cs), cc)


# ---- merger rate as a function of redshift: R(z) = rate * pop_frac * (1 + z)^lamb [* exp(spline(log z))] ----------------
def _rate_factor(z, lamb):
    """(1 + z)^lamb, no normaliser and no dVc/dz (calculations.py:253): the bare power law of the engine's term library on
    the column log(1 + z)."""
    side = side_of(z)
    return Factor(N.TERM_POWERLAW, side, [Column("log1p", z)], [lamb], consts=(0.0, 1.0), flags=N.POWERLAW_UNNORMALISED, tag="rate_of_z")


def calculate_powerlaw_rate_of_z_ppds(lamb, rate, z_model, pop_frac=None):
    """calculations.py:244-258 -> ``(rs, zs)``: ``rs[i] = rate[i] * pop_frac[i] * (1 + zs)^lamb[i]`` on ``z_model.zs``."""
    lamb = np.atleast_1d(np.asarray(lamb, dtype=np.float64))
    n = len(lamb)
    rate, pop_frac = _ones(rate, n), _ones(pop_frac, n)
    zs = np.asarray(z_model.zs, dtype=np.float64)
    p = _curves(zs, lambda z, la: Density([_rate_factor(z, la)], side_of(z)), list(lamb), 1.0)
    return rate[:, None] * pop_frac[:, None] * p, zs


def calculate_powerlaw_spline_rate_of_z_ppds(lamb, z_cs, rate, z_model, pop_frac=None):
    """calculations.py:261-276 -> ``(rs, zs)``: the power law times ``exp(sum_k c_k B_k(log z))`` with the redshift model's
    un-normalised LogXBSpline (spline_perturbation.py:317) and the first coefficient pinned to 0 (:269)."""
    lamb = np.atleast_1d(np.asarray(lamb, dtype=np.float64))
    z_cs = np.atleast_2d(np.asarray(z_cs, dtype=np.float64))
    n = len(lamb)
    rate, pop_frac = _ones(rate, n), _ones(pop_frac, n)
    zs = np.asarray(z_model.zs, dtype=np.float64)
    it = z_model.interpolator
    if z_cs.shape[1] != it.N - 1:
        raise ValueError(f"z_cs must hold {it.N - 1} coefficients per draw (the first of the model's {it.N} is pinned to 0)")

    def dens(z, d):
        side = side_of(z)
        sp = Factor(N.TERM_EXP_SPLINE, side, [Column("log", z)], coefs=d[1], consts=(it.lo, it.hi), n_basis=it.N, flags=N.SPLINE_OUTSIDE_ZERO_EXPONENT, tag="rate_of_z_spline")
        return Density([_rate_factor(z, d[0]), sp], side)

    draws = [(float(lamb[i]), np.concatenate([[0.0], z_cs[i]])) for i in range(n)]
    p = _curves(zs, dens, draws, (1.0, np.zeros(it.N)))
    return rate[:, None] * pop_frac[:, None] * p, zs
