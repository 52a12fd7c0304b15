"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT.

A NumPy (float64) restatement of the reference algorithm for GWInferno's hot path: population
densities at the PE tensor ``(N_ev, N_pe)`` and the found-injection vector ``(N_inj,)``,
importance-sampling reductions, and the assembly of ``log_l`` with its diagnostic sites.  It
deliberately follows the REFERENCE's formulation -- dense ``(N_basis, N_valid)`` design matrices
from the Cox-de Boor recursion, masked scatter, linear-domain products, trapezoid normalisers on
the reference's grids -- and not the engine's (closed-form 4-tap splines, log-domain streaming),
so that agreement between the two is evidence and not tautology.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (gwinferno_amd) never does, and fails loudly without its HIP library.

Parity pin: every function below is checked against golden vectors produced by running the
unmodified reference (tests/golden/make_golden.py -> tests/golden/*.npz; see
tests/test_oracle_golden.py).  Citations are ``file:line`` relative to the reference root.
"""
import numpy as np
from scipy.special import betaln as _betaln
from scipy.special import erf as _erf
from scipy.special import logsumexp as _logsumexp

NEG_BIG = float(np.nan_to_num(-np.inf))  # what jnp.nan_to_num(-inf) yields (analysis.py:280)


# ==========================================================================================
# 0. cosmology tables  (gwinferno/cosmology.py) -- setup-time only
# ==========================================================================================
class _Planck15LVK:
    """cosmology.py:27-155 restated literally: step-by-step trapezoid recurrence for the comoving
    distance on z = arange(0, 10, 1e-3) (``update`` :48-63, ``extend`` :65-77), linear
    interpolation (``z2Dc`` :111-120), dVc/dz = 4 pi Dc^2 (c/H0)/E(z) (:95-101); LVK Planck-2015
    constants (:13, :19-22)."""

    C_SI = 299792458.0
    H0 = 67.90 / 1e-3
    OM = 0.3065

    def __init__(self):
        self.c_over_H0 = self.C_SI / self.H0
        self.OL = 1.0 - self.OM
        self.z = np.arange(0, 10.0, 1e-3)
        dz = self.z[1] - self.z[0]
        dc = np.zeros_like(self.z)
        for i in range(len(self.z) - 1):
            dc[i + 1] = dc[i] + 0.5 * (self.dDc_dz(self.z[i]) + self.dDc_dz(self.z[i] + dz)) * dz
        self.Dc = dc

    def dDc_dz(self, z):
        opz = 1.0 + z
        return self.c_over_H0 / (self.OL + self.OM * opz**3) ** 0.5

    def dVc_dz(self, z):
        z = np.asarray(z, dtype=np.float64)
        return 4 * np.pi * np.interp(z, self.z, self.Dc) ** 2 * self.dDc_dz(z)


_COSMO = None


def planck15_lvk():
    global _COSMO
    if _COSMO is None:
        _COSMO = _Planck15LVK()
    return _COSMO


# ==========================================================================================
# 1. B-spline bases  (gwinferno/interpolation.py)
# ==========================================================================================
def uniform_knot_vector(n_basis, lo, hi, order=4):
    """interpolation.py:98-106 -- ``n_basis - order + 2`` interior knots on [lo, hi], extended by
    ``order - 1`` ghost knots of the same spacing on each side."""
    interior = np.linspace(lo, hi, n_basis - order + 2)
    dx = interior[1] - interior[0]
    knots = np.linspace(lo - dx * (order - 1), hi + dx * (order - 1), len(interior) + 2 * (order - 1))
    assert len(knots) == n_basis + order
    return knots


def cox_de_boor_bsplines(xs, knots, n_basis, order=4):
    """Partition-of-unity B-splines B_i(xs), i < n_basis, by the Cox-de Boor recurrence in the
    reference's M-spline normalisation (interpolation.py:128-149) followed by the
    ``(t_{i+k} - t_i) / k`` rescaling of BSpline._bases (:268-278).  Written bottom-up (one table
    per order) instead of the reference's top-down recursion; same arithmetic per entry.
    Returns ``(n_basis, *xs.shape)``."""
    xs = np.asarray(xs, dtype=np.float64)
    t = np.asarray(knots, dtype=np.float64)
    # order-1 pieces: 1/(t_{i+1}-t_i) on the half-open interval [t_i, t_{i+1})   (:143-146)
    level = []
    for i in range(n_basis + order - 1):
        width = t[i + 1] - t[i]
        if width < 1e-6:  # degenerate guard (:141)
            level.append(np.zeros_like(xs))
        else:
            level.append(np.where((xs >= t[i]) & (xs < t[i + 1]), 1.0 / width, 0.0))
    for k in range(2, order + 1):
        nxt = []
        for i in range(n_basis + order - k):
            span = t[i + k] - t[i]
            if span < 1e-6:
                nxt.append(np.zeros_like(xs))
            else:
                v = (xs - t[i]) * level[i] + (t[i + k] - xs) * level[i + 1]  # (:148)
                nxt.append((v * k) / ((k - 1) * span))  # (:149)
        level = nxt
    out = np.stack([(t[i + order] - t[i]) / order * level[i] for i in range(n_basis)])  # (:278)
    return out


class SplineBasis:
    """The four 1-D bases the population models use (interpolation.py:236-449), selected by
    ``kind``:  'B' (BSpline), 'logX' (LogXBSpline), 'logY' (LogYBSpline), 'logXlogY'
    (LogXLogYBSpline)."""

    GRID_POINTS = {"B": 1000, "logX": 1000, "logY": 1000, "logXlogY": 1500}  # :111, :343, :378, :433

    def __init__(self, kind, n_basis, xrange, normalize, order=4):
        assert kind in self.GRID_POINTS
        self.kind, self.n, self.order, self.normalize = kind, n_basis, order, normalize
        self.log_x = kind in ("logX", "logXlogY")
        self.log_y = kind in ("logY", "logXlogY")
        # log-X bases live on log(xrange) (:337, :427)
        self.lo, self.hi = (np.log(xrange[0]), np.log(xrange[1])) if self.log_x else (xrange[0], xrange[1])
        self.knots = uniform_knot_vector(n_basis, self.lo, self.hi, order)
        if normalize:
            ends = (np.exp(self.lo), np.exp(self.hi)) if self.log_x else (self.lo, self.hi)
            self.grid = np.linspace(ends[0], ends[1], self.GRID_POINTS[kind])
            self.grid_design = self.design(self.grid)

    def design(self, xs):
        """``bases(xs)``: (n_basis, *xs.shape).  Outside the closed domain the column is 0
        (:175) for the linear-Y bases and -inf (:407, :449) for the log-Y bases."""
        xs = np.asarray(xs, dtype=np.float64)
        with np.errstate(all="ignore"):
            coord = np.log(xs) if self.log_x else xs
        dm = cox_de_boor_bsplines(coord, self.knots, self.n, self.order)
        outside = (coord < self.lo) | (coord > self.hi)
        dm = np.where(outside, 0.0, dm)
        if self.log_y:
            dm = np.where(outside, -np.inf, dm)
        return dm

    def contract(self, design, coefs):
        """``_project``: sum_k c_k B_k  (:304); log-Y bases exponentiate with NaN/+inf -> -inf (:393-394)."""
        with np.errstate(all="ignore"):
            lin = np.tensordot(np.asarray(coefs, dtype=np.float64), design, axes=(0, 0))
            if not self.log_y:
                return lin
            lin = np.nan_to_num(lin, nan=-np.inf, posinf=-np.inf)
            return np.exp(lin)

    def norm(self, coefs):
        """1 / trapz(_project(grid_bases, coefs), grid) when normalising, else 1  (:280-291)."""
        if not self.normalize:
            return 1.0
        return 1.0 / np.trapezoid(self.contract(self.grid_design, coefs), self.grid)

    def project(self, design, coefs):
        return self.contract(design, coefs) * self.norm(coefs)  # (:306-317)


# ==========================================================================================
# 2. closed-form densities  (gwinferno/distributions.py)
# ==========================================================================================
def powerlaw_pdf(xx, alpha, low, high, floor=0.0):
    """distributions.py:100-119 (``low`` may be an array, e.g. mmin/m1)."""
    with np.errstate(all="ignore"):
        alpha = np.float64(alpha)
        low = np.asarray(low, dtype=np.float64)
        high = np.float64(high)
        dens = np.power(xx, alpha)
        general = (1 + alpha) / (high ** (1 + alpha) - low ** (1 + alpha))
        logcase = 1 / np.log(high / low)
        dens = dens * np.where(alpha == -1, logcase, general)
        return np.where((xx < low) | (xx > high), floor, dens)


def truncnorm_pdf(xx, mu, sig, low, high):
    """distributions.py:122-143, log=False branch."""
    with np.errstate(all="ignore"):
        root2 = 2**0.5
        gauss = np.exp(-np.power(xx - mu, 2) / (2 * sig**2))
        full_norm = 1 / (sig * (2 * np.pi) ** 0.5)
        cdf_lo = 0.5 * (1 + _erf((low - mu) / (sig * root2)))
        cdf_hi = 0.5 * (1 + _erf((high - mu) / (sig * root2)))
        return np.where((xx > high) | (xx < low), 0.0, gauss * (full_norm / (cdf_hi - cdf_lo)))


def betadist(xx, alpha, beta, scale=1.0, floor=0.0):
    """distributions.py:146-162."""
    with np.errstate(all="ignore"):
        ln = (alpha - 1) * np.log(xx) + (beta - 1) * np.log(scale - xx) - (alpha + beta - 1) * np.log(scale)
        ln = ln - _betaln(alpha, beta)
        return np.where((xx <= scale) & (xx >= 0), np.exp(ln), floor)


# ==========================================================================================
# 3. parametric population models  (gwinferno/models/parametric/parametric.py)
# ==========================================================================================
def smooth(dx, x, xmin):
    """distributions.py:16-21, literally: the second ``where`` is true for every x, so s2 = 1/(func+1) everywhere."""
    with np.errstate(all="ignore"):
        func = np.exp(dx / (x - xmin) + dx / (x - xmin - dx))
        s1 = np.where(x < xmin, 0, 1)
        return np.where((x < xmin + dx) | (x >= xmin), (func + 1) ** (-1), s1)


def plpeak_primary_pdf(m1, alpha, mmin, mmax, mpp, sigpp, lam, delta=None):
    """parametric.py:49-53."""
    if delta is None:
        return (1 - lam) * powerlaw_pdf(m1, alpha, mmin, mmax) + lam * truncnorm_pdf(m1, mpp, sigpp, mmin, mmax)
    return (1 - lam) * powerlaw_pdf(m1, alpha, mmin, mmax) * smooth(delta, m1, mmin) + lam * truncnorm_pdf(m1, mpp, sigpp, mmin, mmax)


def powerlaw_primary_ratio_pdf(m1, q, alpha, beta, mmin, mmax):
    """parametric.py:27-30."""
    return powerlaw_pdf(q, beta, mmin / m1, 1.0) * powerlaw_pdf(m1, alpha, mmin, mmax)


def plpeak_primary_ratio_pdf(m1, q, alpha, beta, mmin, mmax, mpp, sigpp, lam, delta=None):
    """parametric.py:39-46."""
    p = powerlaw_pdf(q, beta, mmin / m1, 1.0) * plpeak_primary_pdf(m1, alpha, mmin, mmax, mpp, sigpp, lam, delta=delta)
    return p if delta is None else p * smooth(delta, q * m1, mmin)


def independent_spin_magnitude_beta_dist(a1, a2, alpha1, beta1, alpha2, beta2):
    """parametric.py:71-81."""
    return betadist(a1, alpha1, beta1) * betadist(a2, alpha2, beta2)


def mixture_isoalign_spin_tilt(ct, xi, sigma):
    """parametric.py:84-86."""
    inside = np.where((ct > 1) | (ct < -1), 0.0, 1.0)
    return inside * (1 - xi) / 2 + xi * truncnorm_pdf(ct, 1.0, sigma, -1.0, 1.0)


def default_spin_tilt(ct1, ct2, xi, sigma):
    """parametric.py:97-102."""
    iso1 = np.where((ct1 > 1) | (ct1 < -1), 0.0, 0.5)
    iso2 = np.where((ct2 > 1) | (ct2 < -1), 0.0, 0.5)
    return (1 - xi) * iso1 * iso2 + xi * truncnorm_pdf(ct1, 1.0, sigma, -1.0, 1.0) * truncnorm_pdf(ct2, 1.0, sigma, -1.0, 1.0)


def independent_spin_tilt(ct1, ct2, xi1, xi2, sig1, sig2):
    """parametric.py:93-94."""
    return mixture_isoalign_spin_tilt(ct1, xi1, sig1) * mixture_isoalign_spin_tilt(ct2, xi2, sig2)


class PowerlawRedshift:
    """parametric.py:112-145.  zmin/zmax from the GLOBAL PE and injection arrays (:114-115);
    1000-point normalisation grid (:116); per-sample dVc/dz tables for injections and PE
    (:117-121) selected by the rank of the ``z`` argument (:139-140)."""

    def __init__(self, z_pe, z_inj):
        cosmo = planck15_lvk()
        self.zmin = max(np.min(z_pe), np.min(z_inj))
        self.zmax = min(np.max(z_pe), np.max(z_inj))
        self.zs = np.linspace(self.zmin, self.zmax, 1000)
        self.dVdz_grid = cosmo.dVc_dz(self.zs)
        self.dVdz_by_rank = {1: cosmo.dVc_dz(z_inj), 2: cosmo.dVc_dz(z_pe)}

    def unnormalised(self, z, dVdz, lamb):
        return dVdz * np.power(1.0 + z, lamb - 1.0)  # (:126-127)

    def normalization(self, lamb):
        return np.trapezoid(self.unnormalised(self.zs, self.dVdz_grid, lamb), self.zs)  # (:123-124)

    def __call__(self, z, lamb):
        dVdz = self.dVdz_by_rank[np.ndim(z)]
        return np.where(z <= self.zmax, self.unnormalised(z, dVdz, lamb) / self.normalization(lamb), 0.0)  # (:141-145)


class PowerlawSplineRedshift(PowerlawRedshift):
    """models/spline_perturbation.py:304-372: power law x exp(B-spline in log z), un-normalised
    LogX basis on (zmin, zmax) (:317), dense design over ALL samples (:318-321)."""

    def __init__(self, n_basis, z_pe, z_inj):
        super().__init__(z_pe, z_inj)
        self.basis = SplineBasis("logX", n_basis, (self.zmin, self.zmax), normalize=False)
        self.design_by_rank = {1: self.basis.design(z_inj), 2: self.basis.design(z_pe)}
        self.grid_design = self.basis.design(self.zs)

    def normalization(self, lamb, cs):
        pz = self.dVdz_grid * np.power(1.0 + self.zs, lamb - 1) * np.exp(self.basis.project(self.grid_design, cs))  # (:334-336)
        return np.trapezoid(pz, self.zs)

    def __call__(self, z, lamb, cs):
        rank = np.ndim(z)
        dens = self.dVdz_by_rank[rank] * np.power(1.0 + z, lamb - 1.0) * np.exp(self.basis.project(self.design_by_rank[rank], cs))  # (:352)
        return np.where(z <= self.zmax, dens / self.normalization(lamb, cs), 0.0)  # (:368-372)


# ==========================================================================================
# 4. 1-D B-spline population models + separable products  (models/bsplines/single.py, separable.py)
# ==========================================================================================
class Spline1D:
    """Base1DBSplineModel (single.py:16-128): validity masks (:54-55), dense design matrices over
    the valid samples only (:56-57), and per-call scatter of the projected values into zeros
    (:90-92, :107-109)."""

    def __init__(self, n_basis, x_pe, x_inj, xrange, kind, normalize=True):
        self.basis = SplineBasis(kind, n_basis, xrange, normalize)
        lo, hi = xrange
        self.ok = {True: (x_pe >= lo) & (x_pe <= hi), False: (x_inj >= lo) & (x_inj <= hi)}
        self.dm = {True: self.basis.design(x_pe[self.ok[True]]), False: self.basis.design(x_inj[self.ok[False]])}

    def __call__(self, coefs, pe_samples=True):
        out = np.zeros(self.ok[pe_samples].shape)
        out[self.ok[pe_samples]] = self.basis.project(self.dm[pe_samples], coefs)
        return out


def spline_mass(n, m_pe, m_inj, mmin, mmax, kind="logXlogY"):
    return Spline1D(n, m_pe, m_inj, (mmin, mmax), kind)  # single.py:358-395


def spline_ratio(n, q_pe, q_inj, qmin, kind="logY"):
    return Spline1D(n, q_pe, q_inj, (qmin, 1), kind)  # single.py:321-355


def spline_spin_magnitude(n, a_pe, a_inj, normalize=True):
    return Spline1D(n, a_pe, a_inj, (0.0, 1.0), "logY", normalize)  # single.py:131-162


def spline_spin_tilt(n, ct_pe, ct_inj, normalize=True):
    return Spline1D(n, ct_pe, ct_inj, (-1.0, 1.0), "logY", normalize)  # single.py:165-196


# ==========================================================================================
# 5. importance-sampling reductions and likelihood assembly  (gwinferno/pipeline/analysis.py)
# ==========================================================================================
def per_event_log_bayes_factors(weights, log=False):
    """analysis.py:50-88."""
    with np.errstate(all="ignore"):
        n_pe = weights.shape[1]
        if log:
            log_bf = _logsumexp(weights, axis=1)
            log_neff = 2 * log_bf - _logsumexp(2 * weights, axis=1)
            log_bf = log_bf - np.log(n_pe)
        else:
            tot = np.sum(weights, axis=1)
            neff = tot**2 / np.sum(weights**2, axis=1)
            log_bf = np.log(tot / n_pe)
            log_neff = np.log(neff)
        return log_bf, log_neff, 1 / np.exp(log_neff) - 1 / n_pe


def detection_efficiency(weights, n_total, log=False):
    """analysis.py:91-136."""
    with np.errstate(all="ignore"):
        if log:
            log_mu = _logsumexp(weights) - np.log(n_total)
            mu = np.exp(log_mu)
            var = np.sum(np.exp(weights) ** 2) / n_total**2 - mu**2 / n_total
        else:
            mu = np.sum(weights) / n_total
            var = np.sum(weights**2) / n_total**2 - mu**2 / n_total
            log_mu = np.log(mu)
        log_neff = 2 * log_mu - np.log(var)
        return log_mu, log_neff, 1 / np.exp(log_neff) - 1 / n_total


def hierarchical_likelihood(
    pe_weights,
    inj_weights,
    total_inj,
    nobs,
    tobs,
    surveyed_hypervolume,
    marginalize_selection=False,
    min_neff_cut=True,
    max_variance_cut=False,
    log=False,
    unscaled_rate=30.0,
):
    """analysis.py:139-319 without the categorical (:246-254) and PPC (:321-355) branches.
    Returns the dict of sites the reference registers with numpyro (names identical)."""
    if max_variance_cut and (marginalize_selection or min_neff_cut):
        raise ValueError("max_variance_cut requires marginalize_selection and min_neff_cut to be False")  # (:237-243)
    s = {}
    with np.errstate(all="ignore"):
        log_bfs, log_neffs, variances = per_event_log_bayes_factors(pe_weights, log=log)
        log_mu, log_neff_inj, var_mu = detection_efficiency(inj_weights, total_inj, log=log)
        s["log_nEff_inj"] = log_neff_inj
        s["log_nEffs"] = log_neffs
        s["logBFs"] = log_bfs
        s["detection_efficiency"] = np.exp(log_mu)
        s["variance_log_BFs"] = variances
        s["variance_log_detection_efficiency"] = var_mu
        s["surveyed_hypervolume"] = surveyed_hypervolume / 1.0e9 * tobs  # (:267)
        s["rate"] = unscaled_rate / np.exp(log_mu) / s["surveyed_hypervolume"]  # (:269)
        if marginalize_selection:
            log_mu = log_mu - (3 + nobs) / (2 * np.exp(log_neff_inj))  # (:271)
        if min_neff_cut:
            log_mu = np.where(log_neff_inj >= np.log(4 * nobs), log_mu, np.inf)  # (:273-277)
        s["selection_factor"] = np.where(np.isinf(log_mu), NEG_BIG, -nobs * log_mu)  # (:278-281)
        s["sum_logBFs"] = np.sum(log_bfs)
        log_l = s["selection_factor"] + s["sum_logBFs"]
        log_l = np.where(np.isnan(log_l), NEG_BIG, np.nan_to_num(log_l))  # (:284-291)
        s["log_l"] = log_l
        if min_neff_cut:
            min_neff = np.exp(np.min(np.nan_to_num(log_neffs)))  # (:295)
            log_l = np.where(min_neff <= nobs, NEG_BIG, log_l)  # (:296-303)
            s["neff_less_Nobs"] = log_l
        s["variance_log_likelihood"] = nobs**2 * var_mu + np.sum(variances)  # (:305-308)
        if max_variance_cut:
            log_l = np.where(s["variance_log_likelihood"] <= 1, log_l, NEG_BIG)  # (:309-317)
            s["variance_less_1"] = log_l
        s["log_likelihood"] = log_l  # numpyro.factor (:319)
    return {k: np.asarray(v, dtype=np.float64) for k, v in s.items()}


def apply_difference_prior(coefs, inv_var, degree=1):
    """models/bsplines/smoothing.py:8-28."""
    d = np.diff(np.asarray(coefs, dtype=np.float64), n=degree)
    return -0.5 * inv_var * np.dot(d, d)


# ==========================================================================================
# 6. model compositions of the BASELINE configs (what a user's NumPyro model function computes)
# ==========================================================================================
def _finite_or_zero(w):
    """tests/inference_test.py:172, 260."""
    return np.where(np.isnan(w) | np.isinf(w), 0.0, w)


class SplineRedshift(Spline1D):
    """BSplineRedshift (single.py:398-492): LogXBSpline on (1e-4, zmax=2.3) by default WITH the basis's own
    normalisation (interpolation.py:320: ``normalize=True``), so ``funcs`` return project() = spline / trapz(spline)
    (:280-317) scattered into zeros; R(z) = exp(funcs) dVc/dz / (1+z) / normalization(cs), where normalization
    integrates exp of the RAW spline (einsum, :471) over a 1000-point grid between the common z bounds (:447-451)."""

    def __init__(self, n_basis, z_pe, z_inj, dVdc_pe, dVdc_inj, zmax=2.3, normalize=True):
        super().__init__(n_basis, z_pe, z_inj, (1e-4, zmax), "logX", normalize=normalize)
        self.zmin = max(np.min(z_pe), np.min(z_inj))
        self.zmax = min(np.max(z_pe), np.max(z_inj))
        self.zgrid = np.linspace(self.zmin, self.zmax, 1000)
        self.dVcdzgrid = planck15_lvk().dVc_dz(self.zgrid)
        self.grid_bases = self.basis.design(self.zgrid)
        self.dV = {True: dVdc_pe, False: dVdc_inj}
        self.z = {True: z_pe, False: z_inj}

    def normalization(self, cs):
        return np.trapezoid(self.dVcdzgrid / (1 + self.zgrid) * np.exp(np.tensordot(np.asarray(cs, dtype=np.float64), self.grid_bases, axes=(0, 0))), self.zgrid)

    def __call__(self, coefs, pe_samples=True):
        return np.exp(super().__call__(coefs, pe_samples)) * self.dV[pe_samples] / (1 + self.z[pe_samples]) / self.normalization(coefs)


class Composition:
    """A population model bound to one catalog: ``weights(params, pe_samples)`` returns the
    linear importance weights p(theta|Lambda)/prior, ``hypervolume(params)`` the redshift
    normaliser passed as ``surveyed_hypervolume``."""

    PARAMS = {}

    def __init__(self, pedict, injdict, mmin=5.0, mmax=100.0):
        self.pe, self.inj, self.mmin, self.mmax = pedict, injdict, float(mmin), float(mmax)

    def data(self, pe_samples):
        return self.pe if pe_samples else self.inj

    def evaluate(self, params, total_inj, tobs=1.0, **flags):
        log = flags.get("log", False)
        with np.errstate(all="ignore"):
            w_pe, w_inj = self.weights(params, True), self.weights(params, False)
            if log:
                w_pe, w_inj = np.log(w_pe), np.log(w_inj)
        nobs = w_pe.shape[0]
        return hierarchical_likelihood(w_pe, w_inj, total_inj, nobs, tobs, self.hypervolume(params), **flags)


class PLTest(Composition):
    """tests/inference_test.py:162-197."""

    PARAMS = {"alpha": (), "beta": (), "lamb": ()}

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.z_model = PowerlawRedshift(pedict["redshift"], injdict["redshift"])

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        with np.errstate(all="ignore"):
            dens = powerlaw_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], p["alpha"], p["beta"], self.mmin, self.mmax)
            return _finite_or_zero(dens * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

    def hypervolume(self, p):
        return self.z_model.normalization(p["lamb"])


class PLPeak(PLTest):
    """BASELINE config 2 (parametric.py:39-53 x :112-145)."""

    PARAMS = {"alpha": (), "beta": (), "mpp": (), "sigpp": (), "lam": (), "lamb": ()}

    def mass_density(self, p, d):
        return plpeak_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], p["alpha"], p["beta"], self.mmin, self.mmax, p["mpp"], p["sigpp"], p["lam"])

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        with np.errstate(all="ignore"):
            return _finite_or_zero(self.mass_density(p, d) * self.z_model(d["redshift"], p["lamb"]) / d["prior"])


class PLPeakFull(PLPeak):
    """BASELINE config 1 (examples/simple_powerlaw_peak_example.py:82-94)."""

    PARAMS = {k: () for k in ("alpha", "beta", "mpp", "sigpp", "lam", "alpha_a1", "beta_a1", "alpha_a2", "beta_a2", "xi1", "xi2", "sig_t1", "sig_t2", "lamb")}

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        with np.errstate(all="ignore"):
            spins = independent_spin_magnitude_beta_dist(d["a_1"], d["a_2"], p["alpha_a1"], p["beta_a1"], p["alpha_a2"], p["beta_a2"])
            tilts = independent_spin_tilt(d["cos_tilt_1"], d["cos_tilt_2"], p["xi1"], p["xi2"], p["sig_t1"], p["sig_t2"])
            return _finite_or_zero(self.mass_density(p, d) * spins * tilts * self.z_model(d["redshift"], p["lamb"]) / d["prior"])


class PLPeakIIDSpins(PLPeak):
    """PL+Peak x iid_spin_magnitude (parametric.py:67-68, amax = 0.9) x iid_spin_tilt (:89-90) x PL z."""

    PARAMS = {k: () for k in ("alpha", "beta", "mpp", "sigpp", "lam", "alpha_a", "beta_a", "xi", "sig_t", "lamb")}
    AMAX = 0.9

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        with np.errstate(all="ignore"):
            spins = betadist(d["a_1"], p["alpha_a"], p["beta_a"], scale=self.AMAX) * betadist(d["a_2"], p["alpha_a"], p["beta_a"], scale=self.AMAX)
            tilts = mixture_isoalign_spin_tilt(d["cos_tilt_1"], p["xi"], p["sig_t"]) * mixture_isoalign_spin_tilt(d["cos_tilt_2"], p["xi"], p["sig_t"])
            return _finite_or_zero(self.mass_density(p, d) * spins * tilts * self.z_model(d["redshift"], p["lamb"]) / d["prior"])


class BSplineTest(Composition):
    """tests/inference_test.py:124-143, 244-285 (BSplinePrimaryBSplineRatio, separable.py:446-530:
    q domain (m2min/mmax, 1) :508)."""

    NM, NQ, NZ = 10, 5, 5

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.PARAMS = {"m1_coefs": (self.NM,), "q_coefs": (self.NQ,), "z_coefs": (self.NZ,), "lamb": ()}
        self.m_model = spline_mass(self.NM, pedict["mass_1"], injdict["mass_1"], self.mmin, self.mmax)
        self.q_model = spline_ratio(self.NQ, pedict["mass_ratio"], injdict["mass_ratio"], self.mmin / self.mmax)
        self.z_model = PowerlawSplineRedshift(self.NZ, pedict["redshift"], injdict["redshift"])

    def spins(self, p, pe_samples):
        return 1.0

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        with np.errstate(all="ignore"):
            mass = self.q_model(p["q_coefs"], pe_samples) * self.m_model(p["m1_coefs"], pe_samples)
            return _finite_or_zero(mass * self.spins(p, pe_samples) * self.z_model(d["redshift"], p["lamb"], p["z_coefs"]) / d["prior"])

    def hypervolume(self, p):
        return self.z_model.normalization(p["lamb"], p["z_coefs"])


class BSplineFull(BSplineTest):
    """BASELINE config 5 (examples/simple_bspline_example.py:47-71; pipeline/utils.py:104-155)."""

    NM, NQ, NA, NT, NZ = 30, 14, 12, 12, 12

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.PARAMS = {"m1_coefs": (self.NM,), "q_coefs": (self.NQ,), "a1_coefs": (self.NA,), "a2_coefs": (self.NA,), "t1_coefs": (self.NT,), "t2_coefs": (self.NT,), "z_coefs": (self.NZ,), "lamb": ()}
        self.a1 = spline_spin_magnitude(self.NA, pedict["a_1"], injdict["a_1"])
        self.a2 = spline_spin_magnitude(self.NA, pedict["a_2"], injdict["a_2"])
        self.t1 = spline_spin_tilt(self.NT, pedict["cos_tilt_1"], injdict["cos_tilt_1"])
        self.t2 = spline_spin_tilt(self.NT, pedict["cos_tilt_2"], injdict["cos_tilt_2"])

    def spins(self, p, pe_samples):
        mags = self.a1(p["a1_coefs"], pe_samples) * self.a2(p["a2_coefs"], pe_samples)  # separable.py:151-153
        tilts = self.t1(p["t1_coefs"], pe_samples) * self.t2(p["t2_coefs"], pe_samples)  # separable.py:290-292
        return mags * tilts


class BSplineDefaults(BSplineFull):
    """The reference's default spline counts (pipeline/utils.py:29-33; IID=False): 165 hyper-parameters."""

    NM, NQ, NA, NT, NZ = 50, 30, 16, 16, 20


class BSplineIID(Composition):
    """BASELINE configs 3/4 (separable.py:295-365, 17-79, 156-218; parametric.py:112-145)."""

    NM, NA, NT = 30, 16, 16

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.PARAMS = {"m1_coefs": (self.NM,), "beta": (), "a_coefs": (self.NA,), "t_coefs": (self.NT,), "lamb": ()}
        self.m_model = spline_mass(self.NM, pedict["mass_1"], injdict["mass_1"], self.mmin, self.mmax)
        self.a1 = spline_spin_magnitude(self.NA, pedict["a_1"], injdict["a_1"])
        self.a2 = spline_spin_magnitude(self.NA, pedict["a_2"], injdict["a_2"])
        self.t1 = spline_spin_tilt(self.NT, pedict["cos_tilt_1"], injdict["cos_tilt_1"])
        self.t2 = spline_spin_tilt(self.NT, pedict["cos_tilt_2"], injdict["cos_tilt_2"])
        self.z_model = PowerlawRedshift(pedict["redshift"], injdict["redshift"])

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        with np.errstate(all="ignore"):
            mass = self.m_model(p["m1_coefs"], pe_samples) * powerlaw_pdf(d["mass_ratio"], p["beta"], self.mmin / d["mass_1"], 1.0)  # separable.py:363-365
            mags = self.a1(p["a_coefs"], pe_samples) * self.a2(p["a_coefs"], pe_samples)  # :77-79
            tilts = self.t1(p["t_coefs"], pe_samples) * self.t2(p["t_coefs"], pe_samples)  # :216-218
            return _finite_or_zero(mass * mags * tilts * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

    def hypervolume(self, p):
        return self.z_model.normalization(p["lamb"])


class PLPeakDefaultTilt(PLPeak):
    """PL+Peak x PL q x default_spin_tilt x PL z."""

    PARAMS = {"alpha": (), "beta": (), "mpp": (), "sigpp": (), "lam": (), "xi": (), "sig_t": (), "lamb": ()}

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        with np.errstate(all="ignore"):
            tilts = default_spin_tilt(d["cos_tilt_1"], d["cos_tilt_2"], p["xi"], p["sig_t"])
            return _finite_or_zero(self.mass_density(p, d) * tilts * self.z_model(d["redshift"], p["lamb"]) / d["prior"])


class BSplineChiEff(Composition):
    """BSplinePrimaryBSplineRatio x BSplineEffectiveSpinDims (linear 'B' bases, normalised: single.py:199-230,
    287-318; separable.py:706-778) x PL z."""

    NM, NQ, NE, NP = 12, 8, 10, 8

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.PARAMS = {"m1_coefs": (self.NM,), "q_coefs": (self.NQ,), "e_coefs": (self.NE,), "p_coefs": (self.NP,), "lamb": ()}
        self.m_model = spline_mass(self.NM, pedict["mass_1"], injdict["mass_1"], self.mmin, self.mmax)
        self.q_model = spline_ratio(self.NQ, pedict["mass_ratio"], injdict["mass_ratio"], self.mmin / self.mmax)
        self.e_model = Spline1D(self.NE, pedict["chi_eff"], injdict["chi_eff"], (-1.0, 1.0), "B", normalize=True)
        self.p_model = Spline1D(self.NP, pedict["chi_p"], injdict["chi_p"], (0.0, 1.0), "B", normalize=True)
        self.z_model = PowerlawRedshift(pedict["redshift"], injdict["redshift"])

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        with np.errstate(all="ignore"):
            mass = self.q_model(p["q_coefs"], pe_samples) * self.m_model(p["m1_coefs"], pe_samples)
            chi = self.e_model(p["e_coefs"], pe_samples) * self.p_model(p["p_coefs"], pe_samples)
            return _finite_or_zero(mass * chi * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

    def hypervolume(self, p):
        return self.z_model.normalization(p["lamb"])


class BSplineComponentMasses(Composition):
    """BSplineIIDComponentMasses (separable.py:533-613) x PL z."""

    NM = 16

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.PARAMS = {"m_coefs": (self.NM,), "beta": (), "lamb": ()}
        self.m1_model = spline_mass(self.NM, pedict["mass_1"], injdict["mass_1"], 3.0, self.mmax)
        self.m2_model = spline_mass(self.NM, pedict["mass_2"], injdict["mass_2"], 3.0, self.mmax)
        with np.errstate(all="ignore"):
            self.q = {True: pedict["mass_2"] / pedict["mass_1"], False: injdict["mass_2"] / injdict["mass_1"]}  # :585
        self.z_model = PowerlawRedshift(pedict["redshift"], injdict["redshift"])

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        q = self.q[pe_samples]
        with np.errstate(all="ignore"):
            pm = self.m1_model(p["m_coefs"], pe_samples) * self.m2_model(p["m_coefs"], pe_samples)
            mass = np.where((q < 0) | (q > 1), 0.0, pm) * np.power(q, p["beta"])  # :609-613
            return _finite_or_zero(mass * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

    def hypervolume(self, p):
        return self.z_model.normalization(p["lamb"])


class BSplineMisc(Composition):
    """PLPeakPrimaryBSplineRatio (separable.py:368-443) x BSplineSymmetricChiEffective (single.py:233-284) x PL z."""

    NQ, NE = 10, 9

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.PARAMS = {"alpha": (), "mpp": (), "sigpp": (), "lam": (), "q_coefs": (self.NQ,), "e_coefs": (self.NE,), "lamb": ()}
        self.q_model = spline_ratio(self.NQ, pedict["mass_ratio"], injdict["mass_ratio"], 0.0)  # BSplineRatio default qmin = 0 (single.py:321-355)
        self.e_model = Spline1D(self.NE, np.abs(pedict["chi_eff"]), np.abs(injdict["chi_eff"]), (0.0, 1.0), "B", normalize=True)  # :257-265
        self.z_model = PowerlawRedshift(pedict["redshift"], injdict["redshift"])

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        with np.errstate(all="ignore"):
            mass = plpeak_primary_pdf(d["mass_1"], p["alpha"], self.mmin, self.mmax, p["mpp"], p["sigpp"], p["lam"]) * self.q_model(p["q_coefs"], pe_samples)
            chi = 0.5 * self.e_model(p["e_coefs"], pe_samples)  # :284
            return _finite_or_zero(mass * chi * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

    def hypervolume(self, p):
        return self.z_model.normalization(p["lamb"])


class BSplineIndependentMasses(Composition):
    """BSplineIndependentComponentMasses (separable.py:616-703: no mask on q) x PL z."""

    N1, N2 = 14, 11

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.PARAMS = {"m1_coefs": (self.N1,), "m2_coefs": (self.N2,), "beta": (), "lamb": ()}
        self.m1_model = spline_mass(self.N1, pedict["mass_1"], injdict["mass_1"], 3.0, self.mmax)
        self.m2_model = spline_mass(self.N2, pedict["mass_2"], injdict["mass_2"], 3.0, self.mmax)
        with np.errstate(all="ignore"):
            self.q = {True: pedict["mass_2"] / pedict["mass_1"], False: injdict["mass_2"] / injdict["mass_1"]}  # :679
        self.z_model = PowerlawRedshift(pedict["redshift"], injdict["redshift"])

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        with np.errstate(all="ignore"):
            mass = self.m1_model(p["m1_coefs"], pe_samples) * self.m2_model(p["m2_coefs"], pe_samples) * np.power(self.q[pe_samples], p["beta"])  # :703
            return _finite_or_zero(mass * self.z_model(d["redshift"], p["lamb"]) / d["prior"])

    def hypervolume(self, p):
        return self.z_model.normalization(p["lamb"])


class BSplineRedshiftCase(Composition):
    """powerlaw_primary_ratio_pdf x BSplineRedshift(8) with the class defaults (single.py:398-492)."""

    NZ, NORMALIZE = 8, True
    PARAMS = {"alpha": (), "beta": (), "z_coefs": (NZ,)}

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        cosmo = planck15_lvk()
        self.z_model = SplineRedshift(self.NZ, pedict["redshift"], injdict["redshift"], cosmo.dVc_dz(pedict["redshift"]), cosmo.dVc_dz(injdict["redshift"]),
                                      normalize=self.NORMALIZE)

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        with np.errstate(all="ignore"):
            dens = powerlaw_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], p["alpha"], p["beta"], self.mmin, self.mmax)
            return _finite_or_zero(dens * self.z_model(p["z_coefs"], pe_samples=pe_samples) / d["prior"])

    def hypervolume(self, p):
        return self.z_model.normalization(p["z_coefs"])


class BSplineRedshiftRawCase(BSplineRedshiftCase):
    NORMALIZE = False


class PLPeakSmooth(PLPeak):
    """plpeak_primary_ratio_pdf with the low-mass taper delta (parametric.py:39-53) x PowerlawRedshiftModel."""

    PARAMS = {"alpha": (), "beta": (), "mpp": (), "sigpp": (), "lam": (), "delta": (), "lamb": ()}

    def weights(self, p, pe_samples):
        d = self.data(pe_samples)
        with np.errstate(all="ignore"):
            dens = plpeak_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], p["alpha"], p["beta"], self.mmin, self.mmax, p["mpp"], p["sigpp"], p["lam"], delta=p["delta"])
            return _finite_or_zero(dens * self.z_model(d["redshift"], p["lamb"]) / d["prior"])


# ==========================================================================================
# log-space distributions summed by construct_hierarchical_model  (gwinferno/numpyro_distributions.py)
# ==========================================================================================
def powerlaw_log_prob(value, alpha, minimum, maximum):
    """Powerlaw.log_prob, numpyro_distributions.py:127-136."""
    with np.errstate(all="ignore"):
        logp = alpha * np.log(value) + np.log((1.0 + alpha) / (maximum ** (1.0 + alpha) - minimum ** (1.0 + alpha)))
        logp_neg1 = -np.log(value) - np.log(maximum / minimum)
        return np.where((value < minimum) | (value > maximum), NEG_BIG, np.where(alpha == -1.0, logp_neg1, logp))


class PowerlawRedshiftDistribution:
    """PowerlawRedshift, numpyro_distributions.py:156-201 (log_prob path only)."""

    def __init__(self, lamb, maximum, zgrid, dVcdz):
        self.lamb, self.maximum, self.zs, self.dVdc = lamb, maximum, np.asarray(zgrid), np.asarray(dVcdz)
        self.norm = np.trapezoid(self.dVdc * (1 + self.zs) ** (lamb - 1), self.zs)  # (:174-175)

    def log_prob(self, value):
        with np.errstate(all="ignore"):
            dV = np.interp(value, self.zs, self.dVdc)  # (:189-190)
            return np.where(value <= self.maximum, np.log(dV) + (self.lamb - 1.0) * np.log(1.0 + value) - np.log(self.norm), NEG_BIG)  # (:191-195)


class BSplineDistribution:
    """numpyro_distributions.py:266-303 (log_prob path only): a gridded log-pdf, linearly interpolated."""

    def __init__(self, cs, grid, grid_dmat):
        self.grid = np.asarray(grid)
        with np.errstate(all="ignore"):
            self.lpdfs = np.nan_to_num(np.einsum("i,i...->...", np.asarray(cs, dtype=np.float64), grid_dmat), nan=-np.inf)  # (:273)
            self.norm = np.trapezoid(np.exp(self.lpdfs), self.grid)  # (:274-275)

    def log_prob(self, value):
        return np.interp(value, self.grid, self.lpdfs) - np.log(self.norm)  # (:296-301)


class ChmPowerlaw(Composition):
    """construct_hierarchical_model (analysis.py:359-424) on Powerlaw mass_1 (sampled bounds) / Powerlaw mass_ratio /
    PowerlawRedshift -- the model of examples/config_files/config.yml.  Always log-space (:422)."""

    ZMAX, QMIN = 1.9, 0.02
    PARAMS = {"alpha": (), "mmin": (), "mmax": (), "beta": (), "lamb": ()}

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.zgrid = np.linspace(1e-9, self.ZMAX, 1000)  # (:371-372)
        self.dV = planck15_lvk().dVc_dz(self.zgrid)

    def log_probs(self, p, d):
        z = PowerlawRedshiftDistribution(p["lamb"], self.ZMAX, self.zgrid, self.dV)
        return [powerlaw_log_prob(d["mass_1"], p["alpha"], p["mmin"], p["mmax"]), powerlaw_log_prob(d["mass_ratio"], p["beta"], self.QMIN, 1.0), z.log_prob(d["redshift"])], z.norm

    def log_weights(self, p, pe_samples):
        d = self.data(pe_samples)
        with np.errstate(all="ignore"):
            lps, _ = self.log_probs(p, d)
            return np.sum(np.array(lps), axis=0) - np.log(d["prior"])  # (:401-402)

    def weights(self, p, pe_samples):
        with np.errstate(all="ignore"):
            return np.exp(self.log_weights(p, pe_samples))

    def hypervolume(self, p):
        return self.log_probs(p, self.inj)[1]  # pop_models["redshift"].norm (:410)

    def evaluate(self, params, total_inj, tobs=1.0, **flags):
        flags = dict(flags, log=True)
        w_pe, w_inj = self.log_weights(params, True), self.log_weights(params, False)
        return hierarchical_likelihood(w_pe, w_inj, total_inj, w_pe.shape[0], tobs, self.hypervolume(params), **flags)


class ChmBSpline(ChmPowerlaw):
    """The same with BSplineDistribution populations for mass_1 (LogXLogYBSpline design matrix on its grid) and
    mass_ratio (LogYBSpline), as tests/numpyro_distributions_test.py:91-129 build them."""

    NM, NQ = 16, 10
    PARAMS = {"m_coefs": (NM,), "q_coefs": (NQ,), "lamb": ()}

    def __init__(self, pedict, injdict, **kw):
        super().__init__(pedict, injdict, **kw)
        self.m_grid = np.linspace(self.mmin, self.mmax, 1000)
        self.m_dmat = SplineBasis("logXlogY", self.NM, (self.mmin, self.mmax), True).design(self.m_grid)
        self.q_grid = np.linspace(0.0, 1.0, 1000)
        self.q_dmat = SplineBasis("logY", self.NQ, (0.0, 1.0), True).design(self.q_grid)

    def log_probs(self, p, d):
        z = PowerlawRedshiftDistribution(p["lamb"], self.ZMAX, self.zgrid, self.dV)
        m = BSplineDistribution(p["m_coefs"], self.m_grid, self.m_dmat)
        q = BSplineDistribution(p["q_coefs"], self.q_grid, self.q_dmat)
        return [m.log_prob(d["mass_1"]), q.log_prob(d["mass_ratio"]), z.log_prob(d["redshift"])], z.norm


COMPOSITIONS = {
    "plpeak_iid_spins": PLPeakIIDSpins,
    "bspline_misc": BSplineMisc,
    "bspline_independent_masses": BSplineIndependentMasses,
    "chm_powerlaw": ChmPowerlaw,
    "chm_bspline": ChmBSpline,
    "plpeak_smooth": PLPeakSmooth,
    "bspline_redshift": BSplineRedshiftCase,
    "bspline_redshift_raw": BSplineRedshiftRawCase,
    "plpeak_default_tilt": PLPeakDefaultTilt,
    "bspline_chieff": BSplineChiEff,
    "bspline_component_masses": BSplineComponentMasses,
    "pl_test": PLTest,
    "plpeak": PLPeak,
    "plpeak_full": PLPeakFull,
    "bspline_test": BSplineTest,
    "bspline_iid": BSplineIID,
    "bspline_full": BSplineFull,
    "bspline_defaults": BSplineDefaults,
}


def fd_gradient(comp, params, total_inj, rel=1e-3, **flags):
    """4th-order central differences of ``log_likelihood`` (the oracle has no autodiff)."""
    def f(q):
        return float(comp.evaluate(q, total_inj, **flags)["log_likelihood"])

    out = {}
    for name, val in params.items():
        arr = np.atleast_1d(np.asarray(val, dtype=np.float64))
        g = np.zeros_like(arr)
        for i in range(arr.size):
            h = rel * max(1.0, abs(arr[i]))
            vals = []
            for k in (-2, -1, 1, 2):
                q = {n: (np.array(v, dtype=np.float64, copy=True) if np.ndim(v) else float(v)) for n, v in params.items()}
                if np.ndim(val):
                    q[name][i] = arr[i] + k * h
                else:
                    q[name] = arr[i] + k * h
                vals.append(f(q))
            g[i] = (vals[0] - 8 * vals[1] + 8 * vals[2] - vals[3]) / (12 * h)
        out[name] = g.reshape(np.shape(val))
    return out
