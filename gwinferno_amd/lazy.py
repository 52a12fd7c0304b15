"""Lazy population densities.

In the reference every model call returns a dense array shaped like the PE tensor ``(N_ev, N_pe)``
or the injection vector ``(N_inj,)`` (e.g. ``Base1DBSplineModel.__call__``,
models/bsplines/single.py:111-128; ``plpeak_primary_ratio_pdf``, models/parametric/parametric.py:39-46),
and the user's model function multiplies those arrays together and divides by the sampling prior
(examples/simple_bspline_example.py:58-71, tests/inference_test.py:168-172).

Here the same calls return a :class:`Density`: a symbolic product of *factors* that remembers which
per-sample data and which hyper-parameters each factor uses.  ``*`` and ``/ prior`` compose them
exactly as the array code does; nothing is evaluated until
:func:`gwinferno_amd.likelihood.hierarchical_likelihood` hands the PE-side and injection-side
products to the HIP engine, which evaluates value, gradient and diagnostics in one fused scan.
"""
import numpy as np

PE, INJ = "pe", "inj"


def side_of(arr):
    """The reference tells PE from injection data by array rank (parametric.py:130-131,
    spline_perturbation.py:351-352): 2-D -> PE samples, 1-D -> injections."""
    nd = np.ndim(arr)
    if nd == 2:
        return PE
    if nd == 1:
        return INJ
    raise ValueError(f"expected a (N_ev, N_pe) or (N_inj,) array, got rank {nd}")


class Column:
    """One per-sample fp64 column for one side: a transform of a user array."""

    __slots__ = ("transform", "source", "const", "_cache")

    def __init__(self, transform, source, const=0.0):
        """``source``: the user's array, or a tuple of two for the product transform; ``const``: the
        subtrahend of "sub" / "prod_sub".  Identity of the SOURCE arrays keys the engine cache."""
        self.transform, self.source, self.const, self._cache = transform, source, float(const), None

    def key(self):
        ids = tuple(id(a) for a in self.source) if isinstance(self.source, tuple) else (id(self.source),)
        return (self.transform, self.const) + ids

    def values(self):
        if self._cache is None:
            if self.transform == "prod_sub":  # a * b - const  (e.g. m2 - mmin = q m1 - mmin)
                a, b = (np.asarray(v, dtype=np.float64) for v in self.source)
                self._cache = np.ascontiguousarray(a * b - self.const)
                return self._cache
            x = np.asarray(self.source, dtype=np.float64)
            with np.errstate(all="ignore"):
                if self.transform == "sub":
                    v = x - self.const
                elif self.transform == "logdiv":    # log(x / const)
                    v = np.log(x / self.const)
                elif self.transform == "log1mdiv":  # log(1 - x / const)
                    v = np.log(1.0 - x / self.const)
                elif self.transform == "id":
                    v = x
                elif self.transform == "log":
                    v = np.log(x)
                elif self.transform == "neglog":
                    v = -np.log(x)
                elif self.transform == "log1m":
                    v = np.log(1.0 - x)
                elif self.transform == "log1p":
                    v = np.log(1.0 + x)
                elif self.transform == "abs":
                    v = np.abs(x)
                else:
                    raise ValueError(self.transform)
            self._cache = np.ascontiguousarray(v, dtype=np.float64)
        return self._cache


class GridNorm:
    """A grid normaliser Z(theta) (see gwi_norm in include/gwi_engine.h).  ``expo_param`` is the index
    (within the owning factor's ``scalars``) of the power-law exponent, ``coefs`` marks that the
    owning factor's spline coefficients enter the integrand -- or ``coefs``, when the integrand uses a
    different coefficient vector from the per-sample term (BSplineRedshift with a normalised basis)."""

    def __init__(self, tw, lb=None, l1=None, expo_param=None, expo_add=0.0, us=None, n_basis=0, lo=0.0, hi=1.0, spline_flags=0, coefs=None):
        self.coefs = coefs
        self.tw = np.ascontiguousarray(tw, dtype=np.float64)
        self.lb = None if lb is None else np.ascontiguousarray(lb, dtype=np.float64)
        self.l1 = None if l1 is None else np.ascontiguousarray(l1, dtype=np.float64)
        self.us = None if us is None else np.ascontiguousarray(us, dtype=np.float64)
        self.expo_param, self.expo_add = expo_param, float(expo_add)
        self.n_basis, self.lo, self.hi, self.spline_flags = int(n_basis), float(lo), float(hi), int(spline_flags)


class Factor:
    """One multiplicative term of a density, bound to ONE side's data."""

    def __init__(self, kind, side, columns, scalars=(), coefs=None, consts=(), n_basis=0, flags=0, mask=None, static_log=None, norm=None,
                 owner=None, norm_owner=None, tag=""):
        self.kind = kind
        self.side = side
        self.columns = list(columns)          # list[Column], order = cols[] of gwi_term
        self.scalars = list(scalars)          # hyper-parameter values, order = theta[] of gwi_term
        self.coefs = coefs                    # spline coefficient vector or None
        self.consts = tuple(float(c) for c in consts)  # p[] of gwi_term
        self.n_basis = int(n_basis)
        self.flags = int(flags)
        self.mask = mask                      # bool array: False -> sample excluded (weight 0)
        self.static_log = static_log          # theta-independent per-sample log factor (e.g. log dVc/dz)
        self.norm = norm                      # GridNorm dividing this factor, or None
        self.owner = owner                    # model object (pairs PE and injection sides)
        self.norm_owner = norm_owner if norm_owner is not None else owner  # normalisers are shared per owner
        self.tag = tag

    def structure(self):
        return (self.kind, self.consts, self.n_basis, self.flags, id(self.owner) if self.owner is not None else None, self.tag)


class Density:
    """Product of factors for one side, optionally divided by the sampling prior and multiplied
    by theta-independent per-sample arrays / constants."""

    def __init__(self, factors=(), side=None, log_static=(), log_const=0.0):
        self.factors = list(factors)
        self.side = side
        self.log_static = list(log_static)  # [(+1 | -1, array)]: theta-independent per-sample factors (x or /), folded into kappa
        self.log_const = log_const          # log of plain scalar multipliers

    # ---- algebra ---------------------------------------------------------------------------------
    def _merge_side(self, other_side):
        if self.side is not None and other_side is not None and self.side != other_side:
            raise ValueError("cannot combine PE-sample and injection densities in one product")
        return self.side if self.side is not None else other_side

    def __mul__(self, other):
        if isinstance(other, Density):
            return Density(self.factors + other.factors, self._merge_side(other.side), self.log_static + other.log_static, self.log_const + other.log_const)
        if np.ndim(other) == 0:
            with np.errstate(all="ignore"):
                return Density(self.factors, self.side, self.log_static, self.log_const + np.log(float(other)))
        return Density(self.factors, self._merge_side(side_of(other)), self.log_static + [(1.0, other)], self.log_const)

    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, Density):
            raise TypeError("division by a lazy density is not supported")
        if np.ndim(other) == 0:
            return Density(self.factors, self.side, self.log_static, self.log_const - np.log(float(other)))
        return Density(self.factors, self._merge_side(side_of(other)), self.log_static + [(-1.0, other)], self.log_const)

    def __repr__(self):
        return f"Density(side={self.side}, factors={[f.kind for f in self.factors]})"


def where_finite(density):
    """Stand-in for ``jnp.where(jnp.isnan(w) | jnp.isinf(w), 0, w)`` (tests/inference_test.py:172, 260):
    the engine always applies that guard, so this is the identity on lazy densities."""
    return density


class LazyNorm:
    """Handle for a normaliser value (e.g. ``z_model.normalization(lamb)``, parametric.py:123-124),
    resolved by the engine to the Z it integrates in the same launch."""

    def __init__(self, owner, scalars=(), coefs=None):
        self.owner, self.scalars, self.coefs = owner, list(scalars), coefs
