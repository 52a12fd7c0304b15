"""Population models with the reference's names, signatures and calling conventions, returning
lazy :class:`~gwinferno_amd.lazy.Density` objects instead of dense arrays.

Mirrors (paths relative to the reference root):
  gwinferno/distributions.py                      powerlaw_pdf, truncnorm_pdf, betadist
  gwinferno/models/parametric/parametric.py       mass / spin functions, PowerlawRedshiftModel
  gwinferno/models/bsplines/single.py             Base1DBSplineModel and its subclasses
  gwinferno/models/bsplines/separable.py          products of 1-D models
  gwinferno/models/spline_perturbation.py         PowerlawSplineRedshiftModel

Construction records the reference's one-time per-sample work (masks, coordinate transforms, dVc/dz per sample) as
setup expressions (gwinferno_amd.expr) which the engine evaluates once, on the device, when it ingests the catalog;
only the small tables (grids, zmin/zmax) are made here in NumPy.  The per-step arithmetic happens in the HIP engine.
"""
import numpy as np

from . import _native as N
from .cosmology import planck15_lvk
from .interpolation import BSpline, LogXBSpline, LogXLogYBSpline, LogYBSpline, trapezoid_weights
from . import expr as E
from .expr import Sym
from .lazy import INJ, PE, Column, Density, Factor, GridNorm, LazyNorm, side_of

__all__ = [
    "powerlaw_pdf", "truncnorm_pdf", "betadist",
    "powerlaw_primary_ratio_pdf", "plpeak_primary_pdf", "plpeak_primary_ratio_pdf",
    "beta_spin_magnitude", "iid_spin_magnitude", "independent_spin_magnitude_beta_dist",
    "mixture_isoalign_spin_tilt", "iid_spin_tilt", "independent_spin_tilt", "default_spin_tilt",
    "BSplineChiEffective", "BSplineSymmetricChiEffective", "BSplineChiPrecess", "BSplineEffectiveSpinDims",
    "BSplineIIDComponentMasses", "BSplineIndependentComponentMasses",
    "PowerlawRedshiftModel", "PowerlawSplineRedshiftModel",
    "Base1DBSplineModel", "BSplineMass", "BSplineRatio", "BSplineSpinMagnitude", "BSplineSpinTilt",
    "BSplineIIDSpinMagnitudes", "BSplineIndependentSpinMagnitudes", "BSplineIIDSpinTilts", "BSplineIndependentSpinTilts",
    "BSplinePrimaryPowerlawRatio", "PLPeakPrimaryBSplineRatio", "BSplinePrimaryBSplineRatio",
]


_CONVERTED = {}  # id(original) -> (original, float64 ndarray)


def _f(x):
    """The caller's array as a float64 ndarray WITHOUT losing its identity across calls: engines are cached on the
    identity of the data arrays, and a model function hands the same objects in on every evaluation.  A plain float64
    ndarray is returned as is, and so is a float32 one (the reference's GWTC-3 tensors: the setup expressions read it as
    fp64, on the device where the engine has one); anything else (memmaps and other ndarray subclasses, lists) is
    converted once and the conversion remembered for as long as the original lives here.  A setup expression passes through."""
    if isinstance(x, Sym) or (type(x) is np.ndarray and x.dtype in (np.float64, np.float32)):
        return x
    if np.ndim(x) == 0:
        return np.asarray(x, dtype=np.float64)
    hit = _CONVERTED.get(id(x))
    if hit is not None and hit[0] is x:
        return hit[1]
    if len(_CONVERTED) > 512:
        _CONVERTED.clear()
    y = np.asarray(x, dtype=np.float64)
    _CONVERTED[id(x)] = (x, y)
    return y


# ================================================================================================
# closed-form densities (gwinferno/distributions.py)
# ================================================================================================
def powerlaw_pdf(xx, alpha, low, high, floor=0.0):
    """distributions.py:100-119.  ``low`` may be a scalar (fixed truncation) or a per-sample array
    (``mmin / m1``, parametric.py:28); the array form requires ``high == 1``."""
    if floor != 0.0:
        raise NotImplementedError("floor != 0 is not used by any reference model")
    xx = _f(xx)
    side = side_of(xx)
    if not isinstance(low, Sym) and np.ndim(low) == 0:
        mask = lambda: ~((Sym.src(xx) < low) | (Sym.src(xx) > high))  # noqa: E731 -- a setup expression, built and evaluated when an engine is bound
        return Density([Factor(N.TERM_POWERLAW, side, [Column("log", xx)], [alpha], consts=(low, high), mask=mask)], side)
    if float(high) != 1.0:
        raise NotImplementedError("per-sample lower bound is implemented for high == 1 (the reference's only use)")
    low = _f(low)
    mask = lambda: ~((Sym.src(xx) < Sym.src(low)) | (Sym.src(xx) > high))  # noqa: E731
    # log r = 0 - (-log low)
    return Density([Factor(N.TERM_POWERLAW_RATIO, side, [Column("log", xx), Column("neglog", low)], [alpha], consts=(0.0,), mask=mask)], side)


def _powerlaw_ratio(q, m1, beta, mmin):
    """powerlaw_pdf(q, beta, mmin/m1, 1) sharing the log m1 column (parametric.py:28, :40; separable.py:364)."""
    q, m1 = _f(q), _f(m1)
    side = side_of(q)
    mask = lambda: ~((Sym.src(q) < mmin / Sym.src(m1)) | (Sym.src(q) > 1))  # noqa: E731 -- built when an engine is bound
    return Density([Factor(N.TERM_POWERLAW_RATIO, side, [Column("log", q), Column("log", m1)], [beta], consts=(np.log(mmin),), mask=mask)], side)


def truncnorm_pdf(xx, mu, sig, low, high, log=False):
    """distributions.py:122-143.  ``log=True`` is the log-normal: a truncated normal in log x on
    [log low, log high] times the Jacobian 1/x (:129-134)."""
    xx = _f(xx)
    side = side_of(xx)
    mask = lambda: ~((Sym.src(xx) > high) | (Sym.src(xx) < low))  # noqa: E731
    if log:
        with np.errstate(all="ignore"):
            return Density([Factor(N.TERM_TRUNCNORM, side, [Column("log", xx)], [mu, sig], consts=(np.log(low), np.log(high)), mask=mask,
                                   static_log=lambda: -E.log(Sym.src(xx)))], side)
    return Density([Factor(N.TERM_TRUNCNORM, side, [Column("id", xx)], [mu, sig], consts=(low, high), mask=mask)], side)


def betadist(xx, alpha, beta, scale=1.0, floor=0.0):
    """distributions.py:146-162.  With ``scale`` = s: Beta in x/s divided by s."""
    if floor != 0.0:
        raise NotImplementedError("floor != 0 is not used by any reference model")
    xx = _f(xx)
    side = side_of(xx)
    scale = float(scale)
    mask = lambda: (Sym.src(xx) <= scale) & (Sym.src(xx) >= 0)  # noqa: E731
    if scale == 1.0:
        return Density([Factor(N.TERM_BETA, side, [Column("log", xx), Column("log1m", xx)], [alpha, beta], mask=mask)], side)
    return Density([Factor(N.TERM_BETA, side, [Column("logdiv", xx, scale), Column("log1mdiv", xx, scale)], [alpha, beta], mask=mask)], side, log_const=-np.log(scale))


# ================================================================================================
# parametric models (gwinferno/models/parametric/parametric.py)
# ================================================================================================
def smooth(dx, x, xmin):
    """distributions.py:16-21.  As written there the second ``where`` holds for every ``x``
    (``x < xmin + dx  |  x >= xmin``), so the function IS ``1 / (1 + exp(dx/(x-xmin) + dx/(x-xmin-dx)))``
    everywhere -- it tends to 1/2, not 1, far above ``xmin + dx``.  Reproduced as is; ``dx`` (delta) is a
    hyper-parameter with an analytic gradient."""
    side = side_of(x)
    return Density([Factor(N.TERM_SMOOTH, side, [Column("sub", x, xmin)], [dx])], side)


def plpeak_primary_pdf(m1, alpha, mmin, mmax, mpp, sigpp, lam, delta=None):
    """parametric.py:49-53; with ``delta`` the power-law component carries the taper ``smooth(delta, m1, mmin)``."""
    m1 = _f(m1)
    side = side_of(m1)
    mask = lambda: ~((Sym.src(m1) < mmin) | (Sym.src(m1) > mmax))  # noqa: E731 -- built when an engine is bound
    if delta is None:  # one column, log m1: the kernel forms m1 = exp(log m1) for the peak (include/gwi_engine.h, GWI_TERM_PLPEAK)
        return Density([Factor(N.TERM_PLPEAK, side, [Column("log", m1)], [alpha, mpp, sigpp, lam], consts=(mmin, mmax), mask=mask)], side)
    cols = [Column("id", m1), Column("log", m1)]
    return Density([Factor(N.TERM_PLPEAK_SMOOTH, side, cols, [alpha, mpp, sigpp, lam, delta], consts=(mmin, mmax), mask=mask)], side)


def powerlaw_primary_ratio_pdf(m1, q, alpha, beta, mmin, mmax):
    """parametric.py:27-30."""
    return _powerlaw_ratio(q, m1, beta, mmin) * powerlaw_pdf(m1, alpha, mmin, mmax)


def plpeak_primary_ratio_pdf(m1, q, alpha, beta, mmin, mmax, mpp, sigpp, lam, delta=None):
    """parametric.py:39-46; with ``delta`` also ``smooth(delta, q m1, mmin)`` on the secondary mass."""
    p = _powerlaw_ratio(q, m1, beta, mmin) * plpeak_primary_pdf(m1, alpha, mmin, mmax, mpp, sigpp, lam, delta=delta)
    if delta is None:
        return p
    side = side_of(q)  # smooth(delta, q * m1, mmin): the product is formed once, keyed by the identity of q and m1
    return p * Density([Factor(N.TERM_SMOOTH, side, [Column("prod_sub", (q, m1), mmin)], [delta])], side)


def beta_spin_magnitude(a, alpha, beta, amax=1):
    return betadist(a, alpha, beta, scale=amax)  # parametric.py:63-64


def iid_spin_magnitude(a1, a2, alpha_mag, beta_mag, amax=1):
    return betadist(a1, alpha_mag, beta_mag, scale=amax) * betadist(a2, alpha_mag, beta_mag, scale=amax)  # :67-68


def independent_spin_magnitude_beta_dist(a1, a2, alpha_mag1, beta_mag1, alpha_mag2, beta_mag2, amax1=1, amax2=1):
    return betadist(a1, alpha_mag1, beta_mag1, scale=amax1) * betadist(a2, alpha_mag2, beta_mag2, scale=amax2)  # :71-81


def mixture_isoalign_spin_tilt(ct, xi_tilt, sigma_tilt):
    """parametric.py:84-86."""
    ct = _f(ct)
    side = side_of(ct)
    mask = lambda: ~((Sym.src(ct) > 1) | (Sym.src(ct) < -1))  # noqa: E731
    return Density([Factor(N.TERM_TILT_MIXTURE, side, [Column("id", ct)], [xi_tilt, sigma_tilt], mask=mask)], side)


def iid_spin_tilt(ct1, ct2, xi_tilt, sigma_tilt):
    return mixture_isoalign_spin_tilt(ct1, xi_tilt, sigma_tilt) * mixture_isoalign_spin_tilt(ct2, xi_tilt, sigma_tilt)  # :89-90


def independent_spin_tilt(ct1, ct2, xi_tilt_1, xi_tilt_2, sigma_tilt1, sigma_tilt2):
    return mixture_isoalign_spin_tilt(ct1, xi_tilt_1, sigma_tilt1) * mixture_isoalign_spin_tilt(ct2, xi_tilt_2, sigma_tilt2)  # :93-94


def default_spin_tilt(ct1, ct2, xi_tilt, sigma_tilt):
    """parametric.py:97-102: (1-xi) iso(ct1) iso(ct2) + xi TN(ct1) TN(ct2), one mixing fraction."""
    ct1, ct2 = _f(ct1), _f(ct2)
    side = side_of(ct1)
    mask = lambda: ~((Sym.src(ct1) > 1) | (Sym.src(ct1) < -1)) & ~((Sym.src(ct2) > 1) | (Sym.src(ct2) < -1))  # noqa: E731
    return Density([Factor(N.TERM_TILT_JOINT, side, [Column("id", ct1), Column("id", ct2)], [xi_tilt, sigma_tilt], mask=mask)], side)


class PowerlawRedshiftModel(object):
    """parametric.py:112-145.  ``zmin``/``zmax`` come from the GLOBAL PE and injection arrays
    (:114-115) -- construct the model before sharding a catalog across GPUs."""

    def __init__(self, z_pe, z_inj):
        cosmo = planck15_lvk()
        z_pe, z_inj = _f(z_pe), _f(z_inj)
        lo_pe, hi_pe, lo_inj, hi_inj = float(np.min(z_pe)), float(np.max(z_pe)), float(np.min(z_inj)), float(np.max(z_inj))
        self.zmin = max(lo_pe, lo_inj)
        self.zmax = min(hi_pe, hi_inj)
        self.zs = np.linspace(self.zmin, self.zmax, 1000)
        self.dVdz_ = cosmo.dVc_dz(self.zs)
        self._z = {PE: z_pe, INJ: z_inj}
        # dVc/dz at every sample (:116): a setup expression -- table interpolation + E(z), evaluated when the catalog is ingested
        self._dVdz = {side: cosmo.dVc_dz_expr(Sym.src(z), max_z=max(hi_pe, hi_inj)) for side, z in self._z.items()}
        self._log_dVdz = {side: E.log(e) for side, e in self._dVdz.items()}
        with np.errstate(all="ignore"):
            self._grid_lb = np.log(self.dVdz_)
            self._grid_l1 = np.log(1.0 + self.zs)
        self._grid_tw = trapezoid_weights(self.zs)

    @property
    def dVdzs(self):
        """``[dVc/dz(z_inj), dVc/dz(z_pe)]`` as arrays (parametric.py:116), computed on first use."""
        if getattr(self, "_dVdzs", None) is None:
            self._dVdzs = [np.asarray(self._dVdz[INJ].numpy()), np.asarray(self._dVdz[PE].numpy())]
        return self._dVdzs

    def _side_data(self, z):
        side = side_of(z)
        z = _f(z)
        if z.shape != self._z[side].shape:
            raise ValueError("z must be the array the model was constructed with (the reference looks its dVc/dz table up by rank, parametric.py:139-140)")
        return side, self._z[side]

    def _powerlaw_factor(self, z, lamb):
        side, zz = self._side_data(z)
        mask = lambda: Sym.src(zz) <= self.zmax  # noqa: E731
        return Factor(N.TERM_POWERLAW_REDSHIFT, side, [Column("log1p", zz)], [lamb], mask=mask, static_log=self._log_dVdz[side], owner=self, tag="plz")

    def normalization(self, lamb):
        return LazyNorm(self, [lamb])

    def __call__(self, z, lamb):
        f = self._powerlaw_factor(z, lamb)
        f.norm = GridNorm(self._grid_tw, lb=self._grid_lb, l1=self._grid_l1, expo_param=(f, 0), expo_add=-1.0)
        return Density([f], f.side)


class PowerlawSplineRedshiftModel(PowerlawRedshiftModel):
    """models/spline_perturbation.py:304-372: power law x exp(B-spline in log z) with an
    un-normalised LogXBSpline on (zmin, zmax) (:317) whose bases are 0 outside the domain."""

    def __init__(self, n_splines, z_pe, z_inj, basis=LogXBSpline):
        super().__init__(z_pe, z_inj)
        if basis is not LogXBSpline:
            raise NotImplementedError("only the default LogXBSpline basis is implemented")
        self.n_splines = int(n_splines)
        self.interpolator = LogXBSpline(self.n_splines, xrange=(self.zmin, self.zmax), k=4, normalize=False)
        self._grid_us = self.interpolator.coordinate(self.zs)

    def normalization(self, lamb, cs):
        return LazyNorm(self, [lamb], cs)

    def __call__(self, z, lamb, cs):
        pl = self._powerlaw_factor(z, lamb)
        it = self.interpolator
        sp = Factor(N.TERM_EXP_SPLINE, pl.side, [Column("log", self._z[pl.side])], coefs=cs, consts=(it.lo, it.hi), n_basis=it.N,
                    flags=N.SPLINE_OUTSIDE_ZERO_EXPONENT, owner=self, tag="zspline")
        sp.norm = GridNorm(self._grid_tw, lb=self._grid_lb, l1=self._grid_l1, expo_param=(pl, 0), expo_add=-1.0, us=self._grid_us, n_basis=it.N, lo=it.lo,
                           hi=it.hi, spline_flags=N.SPLINE_OUTSIDE_ZERO_EXPONENT)
        return Density([pl, sp], pl.side)


# ================================================================================================
# 1-D B-spline models (gwinferno/models/bsplines/single.py)
# ================================================================================================
class Base1DBSplineModel(object):
    """single.py:16-128.  Bases: the exponentiated LogYBSpline / LogXLogYBSpline (defaults of every mass /
    ratio / spin-component model, :151, :185, :344, :384) and the linear BSpline (defaults of the
    effective-spin models, :219, :254, :307)."""

    def __init__(self, n_splines, xx, xx_inj, xrange=(0.0, 1.0), degree=3, basis=BSpline, **kwargs):
        if degree != 3:
            raise NotImplementedError("only cubic splines are implemented")
        if basis not in (LogYBSpline, LogXLogYBSpline, BSpline, LogXBSpline):
            raise NotImplementedError(f"basis {getattr(basis, '__name__', basis)} is not implemented for 1-D density models (the four cubic B-spline bases of interpolation.py are)")
        self._linear = basis in (BSpline, LogXBSpline)  # linear-Y: the density is the spline itself
        self.n_splines = int(n_splines)
        self.xmin, self.xmax = xrange
        self.degree = degree
        self.interpolator = basis(n_splines, xrange=xrange, k=degree + 1, **kwargs)
        it = self.interpolator
        self._x = {PE: _f(xx), INJ: _f(xx_inj)}
        self._mask = {}
        self._coord = {}
        for side, x in self._x.items():
            X = Sym.src(x)
            valid = (X >= self.xmin) & (X <= self.xmax)  # single.py:54-55
            coord = E.log(X) if it.log_x else X
            valid = valid & ~((coord < it.lo) | (coord > it.hi))  # -inf columns of the log-Y bases (:407, :449)
            self._mask[side] = valid
            # excluded samples never reach the spline (kappa = -inf); park them inside the domain
            self._coord[side] = E.where(valid, coord, it.lo)
        self._norm = None
        if it.normalize:
            tw, us = it.grid_tables()
            self._norm = GridNorm(tw, us=us, n_basis=it.N, lo=it.lo, hi=it.hi, spline_flags=N.NORM_LINEAR_SPLINE if self._linear else 0)
        self.scale = 1.0

    def _factor(self, coefs, pe_samples):
        side = PE if pe_samples else INJ
        it = self.interpolator
        kind = N.TERM_LINEAR_SPLINE if self._linear else N.TERM_EXP_SPLINE
        return Factor(kind, side, [Column("id", self._coord[side])], coefs=coefs, consts=(it.lo, it.hi), n_basis=it.N, mask=self._mask[side], norm=self._norm, owner=self)

    def __call__(self, coefs, pe_samples=True):
        side = PE if pe_samples else INJ
        d = Density([self._factor(coefs, pe_samples)], side)
        return d if self.scale == 1.0 else d * self.scale


class BSplineSpinMagnitude(Base1DBSplineModel):
    def __init__(self, n_splines, a, a_inj, basis=LogYBSpline, **kwargs):  # single.py:131-162
        xrange = kwargs.pop("xrange", (0.0, 1.0))
        super().__init__(n_splines, a, a_inj, basis=basis, xrange=xrange, **kwargs)


class BSplineSpinTilt(Base1DBSplineModel):
    def __init__(self, n_splines, ct, ct_inj, basis=LogYBSpline, **kwargs):  # single.py:165-196
        xrange = kwargs.pop("xrange", (-1.0, 1.0))
        super().__init__(n_splines, ct, ct_inj, basis=basis, xrange=xrange, **kwargs)


class BSplineChiEffective(Base1DBSplineModel):
    def __init__(self, n_splines, chieff, chieff_inj, basis=BSpline, **kwargs):  # single.py:199-230
        xrange = kwargs.pop("xrange", (-1.0, 1.0))
        super().__init__(n_splines, chieff, chieff_inj, basis=basis, xrange=xrange, **kwargs)


class BSplineSymmetricChiEffective(Base1DBSplineModel):
    def __init__(self, n_splines, chieff, chieff_inj, basis=BSpline, **kwargs):  # single.py:233-284
        xrange = kwargs.pop("xrange", (0.0, 1.0))
        super().__init__(n_splines, abs(Sym.src(_f(chieff))), abs(Sym.src(_f(chieff_inj))), basis=basis, xrange=xrange, **kwargs)
        self.scale = 0.5  # :284


class BSplineChiPrecess(Base1DBSplineModel):
    def __init__(self, n_splines, chip, chip_inj, basis=BSpline, **kwargs):  # single.py:287-318
        xrange = kwargs.pop("xrange", (0.0, 1.0))
        super().__init__(n_splines, chip, chip_inj, basis=basis, xrange=xrange, **kwargs)


class BSplineRatio(Base1DBSplineModel):
    def __init__(self, n_splines, q, q_inj, qmin=0, basis=LogYBSpline, **kwargs):  # single.py:321-355
        xrange = kwargs.pop("xrange", (qmin, 1))
        super().__init__(n_splines, q, q_inj, basis=basis, xrange=xrange, **kwargs)


class BSplineMass(Base1DBSplineModel):
    def __init__(self, n_splines, m, m_inj, mmin=2, mmax=100, basis=LogXLogYBSpline, **kwargs):  # single.py:358-395
        xrange = kwargs.pop("xrange", (mmin, mmax))
        super().__init__(n_splines, m, m_inj, basis=basis, xrange=xrange, **kwargs)


class BSplineRedshift(object):
    """single.py:398-492: ``exp(f(z)) dVc/dz / (1+z) / normalization(coefs)`` with ``f`` a LogXBSpline on
    ``xrange = (1e-4, zmax)``, exponent 0 outside that range (the masked scatter of :77-109), normalised
    by ``trapz(dVc/dz/(1+z) exp(sum c_k B_k), zgrid)`` on a 1000-point grid between the common redshift
    bounds of the PE and injection sets (:447-451, :454-472).

    The reference's default basis is LogXBSpline with ``normalize=True``: ``funcs`` then evaluate
    ``project`` = ``sum c_k B_k / trapz(sum c_k B_k(grid), grid)`` (interpolation.py:280-317), i.e. the
    per-sample exponent uses the coefficients scaled by ``1 / (c . I)``, ``I_k = trapz(B_k(grid), grid)``,
    while ``normalization`` uses the raw ones (:471).  That scaling is theta-only, so it is applied here,
    on the host, to the coefficient vector handed to the engine (``c / (c @ I)`` also traces under JAX,
    whose autodiff then supplies the chain rule); the grid normaliser gets the raw vector."""

    def __init__(self, n_splines, z, z_inj, dVdc, dVdc_inj, zmax=2.3, basis=LogXBSpline, **kwargs):
        if basis is not LogXBSpline:
            raise NotImplementedError("only the default LogXBSpline basis is implemented")
        xrange = kwargs.pop("xrange", (1e-4, zmax))
        degree = kwargs.pop("degree", 3)
        if degree != 3:
            raise NotImplementedError("only cubic splines are implemented")
        kwargs.setdefault("normalize", True)  # LogXBSpline's default (interpolation.py:320)
        self.n_splines = int(n_splines)
        self.xmin, self.xmax = xrange
        self.interpolator = it = LogXBSpline(self.n_splines, xrange=xrange, k=4, **kwargs)
        z, z_inj = _f(z), _f(z_inj)
        self._z = {PE: z, INJ: z_inj}
        self.zmin = max(float(np.min(z)), float(np.min(z_inj)))
        self.zmax = min(float(np.max(z)), float(np.max(z_inj)))
        self.zgrid = np.linspace(self.zmin, self.zmax, 1000)
        self.dVcdzgrid = planck15_lvk().dVc_dz(self.zgrid)
        self._static = {PE: E.log(Sym.src(_f(dVdc))) - E.log1p(Sym.src(z)), INJ: E.log(Sym.src(_f(dVdc_inj))) - E.log1p(Sym.src(z_inj))}
        with np.errstate(all="ignore"):
            self._grid_lb = np.log(self.dVcdzgrid) - np.log1p(self.zgrid)
        self._grid_tw = trapezoid_weights(self.zgrid)
        self._grid_us = it.coordinate(self.zgrid)
        # integrals of the basis functions over the basis's own normalisation grid (linear in z, :343)
        self._basis_integrals = it.bases(it.grid) @ trapezoid_weights(it.grid) if it.normalize else None

    def _exponent_coefs(self, coefs):
        if self._basis_integrals is None:
            return coefs
        return coefs / (coefs @ self._basis_integrals)

    def normalization(self, cs):
        return LazyNorm(self, [], cs)

    def __call__(self, coefs, pe_samples=True):
        side = PE if pe_samples else INJ
        it = self.interpolator
        f = Factor(N.TERM_EXP_SPLINE, side, [Column("log", self._z[side])], coefs=self._exponent_coefs(coefs), consts=(it.lo, it.hi), n_basis=it.N,
                   flags=N.SPLINE_OUTSIDE_ZERO_EXPONENT, static_log=self._static[side], owner=self, tag="zspline")
        f.norm = GridNorm(self._grid_tw, lb=self._grid_lb, us=self._grid_us, n_basis=it.N, lo=it.lo, hi=it.hi, spline_flags=N.SPLINE_OUTSIDE_ZERO_EXPONENT,
                          coefs=coefs if it.normalize else None)
        return Density([f], side)


# ================================================================================================
# separable products (gwinferno/models/bsplines/separable.py)
# ================================================================================================
class _PairModel(object):
    primary_cls = secondary_cls = None

    def _pair(self, c1, c2, pe_samples):
        return self.primary_model(c1, pe_samples=pe_samples) * self.secondary_model(c2, pe_samples=pe_samples)


class BSplineIIDSpinMagnitudes(_PairModel):
    def __init__(self, n_splines, a1, a2, a1_inj, a2_inj, **kwargs):  # separable.py:17-79
        self.primary_model = BSplineSpinMagnitude(n_splines=n_splines, a=a1, a_inj=a1_inj, **kwargs)
        self.secondary_model = BSplineSpinMagnitude(n_splines=n_splines, a=a2, a_inj=a2_inj, **kwargs)

    def __call__(self, coefs, pe_samples=True):
        return self._pair(coefs, coefs, pe_samples)


class BSplineIndependentSpinMagnitudes(_PairModel):
    def __init__(self, n_splines1, n_splines2, a1, a2, a1_inj, a2_inj, kwargs1={}, kwargs2={}, **kwargs):  # separable.py:82-153
        self.primary_model = BSplineSpinMagnitude(n_splines=n_splines1, a=a1, a_inj=a1_inj, **kwargs1, **kwargs)
        self.secondary_model = BSplineSpinMagnitude(n_splines=n_splines2, a=a2, a_inj=a2_inj, **kwargs2, **kwargs)

    def __call__(self, pcoefs, scoefs, pe_samples=True):
        return self._pair(pcoefs, scoefs, pe_samples)


class BSplineIIDSpinTilts(_PairModel):
    def __init__(self, n_splines, ct1, ct2, ct1_inj, ct2_inj, **kwargs):  # separable.py:156-218
        self.primary_model = BSplineSpinTilt(n_splines=n_splines, ct=ct1, ct_inj=ct1_inj, **kwargs)
        self.secondary_model = BSplineSpinTilt(n_splines=n_splines, ct=ct2, ct_inj=ct2_inj, **kwargs)

    def __call__(self, coefs, pe_samples=True):
        return self._pair(coefs, coefs, pe_samples)


class BSplineIndependentSpinTilts(_PairModel):
    def __init__(self, n_splines1, n_splines2, ct1, ct2, ct1_inj, ct2_inj, kwargs1={}, kwargs2={}, **kwargs):  # separable.py:221-292
        self.primary_model = BSplineSpinTilt(n_splines=n_splines1, ct=ct1, ct_inj=ct1_inj, **kwargs1, **kwargs)
        self.secondary_model = BSplineSpinTilt(n_splines=n_splines2, ct=ct2, ct_inj=ct2_inj, **kwargs2, **kwargs)

    def __call__(self, pcoefs, scoefs, pe_samples=True):
        return self._pair(pcoefs, scoefs, pe_samples)


class BSplineEffectiveSpinDims(object):
    def __init__(self, n_splines_e, n_splines_p, chieff, chip, chieff_inj, chip_inj, kwargs_e={}, kwargs_p={}, **kwargs):  # separable.py:706-778
        self.chi_eff_model = BSplineChiEffective(n_splines_e, chieff, chieff_inj, **kwargs_e, **kwargs)
        self.chi_p_model = BSplineChiPrecess(n_splines_p, chip, chip_inj, **kwargs_p, **kwargs)

    def __call__(self, ecoefs, pcoefs, pe_samples=True):
        return self.chi_eff_model(ecoefs, pe_samples=pe_samples) * self.chi_p_model(pcoefs, pe_samples=pe_samples)


class _ComponentMasses(object):
    """p(m1) p(m2) (m2/m1)^beta (separable.py:533-703)."""

    mask_ratio = False

    def _setup(self, m1, m2, m1_inj, m2_inj):
        self._q = [Sym.src(_f(m2_inj)) / Sym.src(_f(m1_inj)), Sym.src(_f(m2)) / Sym.src(_f(m1))]  # :585, :679

    @property
    def qs(self):
        """``[m2_inj / m1_inj, m2 / m1]`` as arrays, computed on first use."""
        if getattr(self, "_qs", None) is None:
            self._qs = [np.asarray(q.numpy()) for q in self._q]
        return self._qs

    def _pairing(self, beta, pe_samples):
        q = self._q[1 if pe_samples else 0]
        side = PE if pe_samples else INJ
        mask = None
        if self.mask_ratio:
            mask = lambda: ~((q < 0) | (q > 1))  # noqa: E731  (:609-613)
        return Density([Factor(N.TERM_POWERLAW, side, [Column("log", q)], [beta], consts=(0.0, 1.0), flags=N.POWERLAW_UNNORMALISED, mask=mask, owner=self, tag="pairing")], side)


class BSplineIIDComponentMasses(_ComponentMasses):
    mask_ratio = True

    def __init__(self, n_splines, m1, m2, m1_inj, m2_inj, mmin=2, mmax=100, **kwargs):  # separable.py:533-613
        self.primary_model = BSplineMass(n_splines=n_splines, m=m1, m_inj=m1_inj, mmin=mmin, mmax=mmax, **kwargs)
        self.secondary_model = BSplineMass(n_splines=n_splines, m=m2, m_inj=m2_inj, mmin=mmin, mmax=mmax, **kwargs)
        self._setup(m1, m2, m1_inj, m2_inj)

    def __call__(self, coefs, beta=0, pe_samples=True):
        return self.primary_model(coefs, pe_samples=pe_samples) * self.secondary_model(coefs, pe_samples=pe_samples) * self._pairing(beta, pe_samples)


class BSplineIndependentComponentMasses(_ComponentMasses):
    def __init__(self, n_splines1, n_splines2, m1, m2, m1_inj, m2_inj, mmin1=2, mmax1=100, mmin2=2, mmax2=100, kwargs1={}, kwargs2={}, **kwargs):
        # separable.py:616-703
        self.primary_model = BSplineMass(n_splines=n_splines1, m=m1, m_inj=m1_inj, mmin=mmin1, mmax=mmax1, **kwargs1, **kwargs)
        self.secondary_model = BSplineMass(n_splines=n_splines2, m=m2, m_inj=m2_inj, mmin=mmin2, mmax=mmax2, **kwargs2, **kwargs)
        self._setup(m1, m2, m1_inj, m2_inj)

    def __call__(self, pcoefs, scoefs, beta=0, pe_samples=True):
        return self.primary_model(pcoefs, pe_samples=pe_samples) * self.secondary_model(scoefs, pe_samples=pe_samples) * self._pairing(beta, pe_samples)


class BSplinePrimaryPowerlawRatio(object):
    def __init__(self, n_splines, m1, m1_inj, mmin=2, mmax=100, **kwargs):  # separable.py:295-365
        self.primary_model = BSplineMass(n_splines, m1, m1_inj, mmin=mmin, mmax=mmax, **kwargs)

    def __call__(self, m1, q, beta, mmin, coefs, pe_samples=True):
        p_m1 = self.primary_model(coefs, pe_samples=pe_samples)
        return p_m1 * _powerlaw_ratio(q, m1, beta, mmin)  # separable.py:363-365


class PLPeakPrimaryBSplineRatio(object):
    def __init__(self, n_splines, q, q_inj, **kwargs):  # separable.py:368-443
        self.ratio_model = BSplineRatio(n_splines, q, q_inj, **kwargs)

    def __call__(self, m1, alpha, mmin, mmax, peak_mean, peak_sd, peak_frac, coefs, pe_samples=True):
        p_q = self.ratio_model(coefs, pe_samples=pe_samples)
        return plpeak_primary_pdf(m1, alpha, mmin, mmax, peak_mean, peak_sd, peak_frac) * p_q  # separable.py:441-443


class BSplinePrimaryBSplineRatio(object):
    def __init__(self, n_splines_m, n_splines_q, m1, m1_inj, q, q_inj, mmax=100.0, m1min=3.0, m2min=3.0, kwargs_m={}, kwargs_q={}, **kwargs):
        # separable.py:446-530; q domain (m2min/mmax, 1) :508
        self.primary_model = BSplineMass(n_splines_m, m1, m1_inj, mmin=m1min, mmax=mmax, **kwargs_m, **kwargs)
        self.ratio_model = BSplineRatio(n_splines_q, q, q_inj, qmin=m2min / mmax, **kwargs_q, **kwargs)

    def __call__(self, mcoefs, qcoefs, pe_samples=True):
        return self.primary_model(mcoefs, pe_samples=pe_samples) * self.ratio_model(qcoefs, pe_samples=pe_samples)
