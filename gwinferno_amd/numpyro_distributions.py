"""``log_prob`` faces of the reference's NumPyro distributions that sit on the likelihood path
(gwinferno/numpyro_distributions.py): what ``construct_hierarchical_model`` sums
(pipeline/analysis.py:401-402) and what the log-space model functions add up
(examples/config_files/model.py:17-22).  ``log_prob(value)`` returns a lazy :class:`~gwinferno_amd.lazy.LogDensity`
for ``value``'s sample set: ``+`` between them and ``- log(prior)`` compose exactly as the reference's array code
does (``*`` and ``/ prior`` of the linear-space models work on them too).  Sampling / cdf / icdf are not on the
log-prob path and are not provided.  :class:`PSplineCoeficientPrior` (numpyro_distributions.py:302-325) acts on
hyper-parameters only and is mirrored whole.

A distribution object is rebuilt on every model call (analysis.py:381-399).  Whatever identifies the kernel and
the device-resident data -- grids, design-matrix descriptors -- is therefore keyed on the identity of the arrays
the caller passes in (kept constant across calls, as ``prior_dict`` constants are), never on the object itself.
"""
import numpy as np

from . import _native as N
from . import expr as E
from .interpolation import DesignMatrix, trapezoid_weights
from .lazy import Column, Factor, GridNorm, LazyNorm, LogDensity, side_of
from .models import _f


class _Token:
    """Stable stand-in for "the same distribution as last call": pairs the PE and injection factors of one
    distribution and names its normaliser.  Holds the keyed arrays so that their ids stay unique."""

    __slots__ = ("key", "keep")

    def __init__(self, key, keep):
        self.key, self.keep = key, keep


_TOKENS = {}


def _token(tag, arrays, consts=()):
    key = (tag,) + tuple(id(a) for a in arrays) + tuple(consts)
    tok = _TOKENS.get(key)
    if tok is None:
        tok = _TOKENS[key] = _Token(key, arrays)
    return tok


class Powerlaw:
    """numpyro_distributions.py:101-153: x^alpha on [minimum, maximum]; the bounds may be sampled
    hyper-parameters (examples/config_files/config.yml:8-25) and are treated as such: one engine serves every
    (alpha, minimum, maximum)."""

    def __init__(self, alpha, minimum=0.0, maximum=1.0, low=0.0, high=1.0, validate_args=None):
        self.alpha, self.minimum, self.maximum = alpha, minimum, maximum

    def log_prob(self, value):
        value = _f(value)
        side = side_of(value)
        # :131-136: -inf (as nan_to_num) outside [minimum, maximum], the alpha = -1 limit handled by the engine
        return LogDensity([Factor(N.TERM_POWERLAW_BOUNDS, side, [Column("log", value), Column("id", value)], [self.alpha, self.minimum, self.maximum])], side)


class PowerlawRedshift:
    """numpyro_distributions.py:156-201: dVc/dz (1+z)^(lamb-1) / trapz(...) on a caller-supplied grid;
    dVc/dz at the samples by linear interpolation into that grid (:189-190).  ``maximum`` is a fixed number
    here (it is in every reference configuration: ``maximum: value: 2.3``)."""

    def __init__(self, lamb, maximum, zgrid, dVcdz, low=0.0, high=1000.0, validate_args=None):
        self.lamb, self.maximum = lamb, float(maximum)
        self.zs = np.asarray(zgrid, dtype=np.float64)
        self.dVdc_ = np.asarray(dVcdz, dtype=np.float64)
        self._owner = _token("PowerlawRedshift", (zgrid, dVcdz), (self.maximum,))
        if not isinstance(self._owner.keep, dict):
            with np.errstate(all="ignore"):
                self._owner.keep = dict(arrays=self._owner.keep, tw=trapezoid_weights(self.zs), lb=np.log(self.dVdc_), l1=np.log(1.0 + self.zs), zs=np.ascontiguousarray(self.zs), dv=np.ascontiguousarray(self.dVdc_))

    @property
    def norm(self):
        """Lazy handle for the normaliser (``surveyed_hypervolume=pop_models["redshift"].norm``, analysis.py:410)."""
        return LazyNorm(self._owner, [self.lamb])

    def log_prob(self, value, dVdc=None):
        value = _f(value)
        side = side_of(value)
        t = self._owner.keep
        Z = E.Sym.src(value)
        mask = Z <= self.maximum
        # per-sample dVc/dz: a setup expression (table interpolation), evaluated once when the catalog is ingested
        logdv = E.log(E.interp(Z, t["zs"], t["dv"])) if dVdc is None else E.log(E.Sym.src(_f(dVdc)))
        f = Factor(N.TERM_POWERLAW_REDSHIFT, side, [Column("log1p", value)], [self.lamb], mask=mask, static_log=logdv, owner=self._owner, tag="plz")
        f.norm = GridNorm(t["tw"], lb=t["lb"], l1=t["l1"], expo_param=(f, 0), expo_add=-1.0)
        return LogDensity([f], side)


class BSplineDistribution:
    """numpyro_distributions.py:266-303: ``lpdfs = cs . grid_dmat`` on ``grid``, ``log_prob(value) =
    interp(value, grid, lpdfs) - log trapz(exp(lpdfs), grid)`` -- a log-density tabulated on the grid and linearly
    interpolated between grid points (end values held outside the grid, as ``jnp.interp`` does).

    ``grid_dmat`` must be ``basis.bases(grid)`` of a basis from :mod:`gwinferno_amd.interpolation` (the same
    call as in the reference, tests/numpyro_distributions_test.py:91-129): the engine recomputes the four taps
    per grid point from the basis description instead of reading a dense matrix."""

    def __init__(self, minimum, maximum, cs, grid, grid_dmat, validate_args=None):
        if not isinstance(grid_dmat, DesignMatrix) or grid_dmat.basis is None:
            raise TypeError("grid_dmat must be basis.bases(grid) from gwinferno_amd.interpolation (a dense matrix of unknown origin cannot be evaluated on the device)")
        basis = grid_dmat.basis
        g = np.asarray(grid, dtype=np.float64)
        self.minimum, self.maximum, self.cs, self.grid, self.basis = minimum, maximum, cs, g, basis
        self._owner = _token("BSplineDistribution", (grid, grid_dmat))
        if not isinstance(self._owner.keep, dict):  # first time these two arrays are seen: check them, build the tables
            if g.ndim != 1 or g.shape != np.shape(grid_dmat.xs) or not np.array_equal(g, grid_dmat.xs):
                _TOKENS.pop(self._owner.key, None)
                raise ValueError("grid is not the grid the design matrix was evaluated on")
            if g.size < 2 or np.any(np.diff(g) <= 0):
                _TOKENS.pop(self._owner.key, None)
                raise ValueError("grid must be strictly increasing")
            us = basis.coordinate(g)
            with np.errstate(all="ignore"):
                outside = ~np.isfinite(us) | basis.outside(us)
            tw = trapezoid_weights(g)
            if basis.log_y:  # lpdf = -inf there: exp -> 0 in the normaliser
                tw = np.where(outside, 0.0, tw)
            # grid points outside the spline domain get a finite coordinate that is recognisably outside
            self._owner.keep = dict(arrays=self._owner.keep, tw=tw, us=np.where(outside, basis.lo - 1.0, us), grid=g)

    def _factor(self, src):
        basis, t = self.basis, self._owner.keep
        value = _f(src)
        side = side_of(value)
        flags = 0 if basis.log_y else N.SPLINE_OUTSIDE_ZERO_EXPONENT
        f = Factor(N.TERM_EXP_SPLINE_LERP, side, [Column("gridindex", value, aux=t["grid"])], coefs=self.cs, consts=(basis.lo, basis.hi), n_basis=basis.N,
                   flags=flags, owner=self._owner, tag="lerp")
        f.norm = GridNorm(t["tw"], us=t["us"], n_basis=basis.N, lo=basis.lo, hi=basis.hi, spline_flags=flags)
        return f, side

    @property
    def norm(self):
        return LazyNorm(self._owner, coefs=self.cs)

    def log_prob(self, value):
        f, side = self._factor(value)
        return LogDensity([f], side)


class PSplineCoeficientPrior:
    """numpyro_distributions.py:302-325: the P-spline smoothing prior as a distribution over a coefficient vector of
    length ``N``: ``log_prob(value) = apply_difference_prior(value, inv_var, diff_order)`` (models/bsplines/smoothing.py:8-28).
    It touches hyper-parameters only -- never sample data -- so it stays host / JAX arithmetic: ``value`` may be a NumPy
    array or a JAX tracer.  Where NumPyro is installed the object is a ``numpyro.distributions.Distribution`` (same
    ``arg_constraints``, real-vector support, event shape ``(N,)``), so ``numpyro.sample(name, PSplineCoeficientPrior(N, tau))``
    works as it does with the reference's class; without NumPyro it is a plain object with the same ``log_prob`` / ``sample``.
    ``sample`` returns ones of shape ``sample_shape + batch_shape`` exactly as the reference's does (:316-318: a placeholder,
    the prior is improper)."""

    def __new__(cls, N, inv_var, diff_order=2, validate_args=None):
        impl = _pspline_numpyro_class()
        if impl is not None and cls is PSplineCoeficientPrior:
            return impl(N, inv_var, diff_order=diff_order, validate_args=validate_args)
        return object.__new__(cls)

    def __init__(self, N, inv_var, diff_order=2, validate_args=None):
        self.N, self.inv_var, self.diff_order = int(N), inv_var, int(diff_order)
        self.batch_shape, self.event_shape = tuple(np.shape(inv_var)), (self.N,)

    def sample(self, key=None, sample_shape=()):
        return np.ones(tuple(sample_shape) + self.batch_shape)

    def log_prob(self, value):
        from .smoothing import apply_difference_prior

        assert tuple(np.shape(value)) == (self.N,)
        return apply_difference_prior(value, self.inv_var, self.diff_order)


_PSPLINE_IMPL = []


def _pspline_numpyro_class():
    """The NumPyro-backed class, built once where numpyro imports (None elsewhere)."""
    if _PSPLINE_IMPL:
        return _PSPLINE_IMPL[0]
    try:
        import jax.numpy as jnp
        from jax import lax
        from numpyro.distributions import Distribution, constraints
        from numpyro.distributions.util import is_prng_key, promote_shapes, validate_sample
    except Exception:
        _PSPLINE_IMPL.append(None)
        return None
    from .smoothing import apply_difference_prior

    class _NumpyroPSplineCoeficientPrior(Distribution):
        arg_constraints = {"inv_var": constraints.positive}
        reparametrized_params = ["inv_var"]

        def __init__(self, N, inv_var, diff_order=2, validate_args=None):
            (self.inv_var,) = promote_shapes(inv_var)
            self._support = constraints.real_vector
            super().__init__(batch_shape=lax.broadcast_shapes(jnp.shape(inv_var)), validate_args=validate_args, event_shape=(N,))
            self.diff_order, self.N = diff_order, N

        @constraints.dependent_property(is_discrete=False, event_dim=0)
        def support(self):
            return self._support

        def sample(self, key, sample_shape=()):
            assert is_prng_key(key)
            return jnp.ones(shape=sample_shape + self.batch_shape)

        @validate_sample
        def log_prob(self, value):
            assert value.shape == (self.N,)
            return apply_difference_prior(value, self.inv_var, self.diff_order)

    _NumpyroPSplineCoeficientPrior.__name__ = _NumpyroPSplineCoeficientPrior.__qualname__ = "PSplineCoeficientPrior"
    _PSPLINE_IMPL.append(_NumpyroPSplineCoeficientPrior)
    return _NumpyroPSplineCoeficientPrior
