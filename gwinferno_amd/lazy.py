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

from . import expr as E

PE, INJ = "pe", "inj"


def side_of(arr):
    """The reference tells PE from injection data by array rank (parametric.py:130-131,
    spline_perturbation.py:351-352): 2-D -> PE samples, 1-D -> injections."""
    nd = getattr(arr, "ndim", None)  # ndarray and Sym answer directly; np.ndim() is for lists
    if nd is None:
        nd = np.ndim(arr)
    if nd == 2:
        return PE
    if nd == 1:
        return INJ
    raise ValueError(f"expected a (N_ev, N_pe) or (N_inj,) array, got rank {nd}")


def _is_number(x):
    """A plain multiplier (Python or NumPy scalar, 0-d array) as opposed to per-sample data (array or setup expression)."""
    nd = getattr(x, "ndim", None)  # ndarray, NumPy scalars and Sym answer directly; np.ndim() is for lists
    if nd is None:
        nd = np.ndim(x)
    return nd == 0 and not isinstance(x, E.Sym)


class Column:
    """One per-sample fp64 column for one side: a transform of a user array, or a whole setup expression."""

    __slots__ = ("transform", "source", "const", "aux", "_cache", "_expr")

    def __init__(self, transform, source, const=0.0, aux=None):
        """``source``: the user's array (or a :class:`~gwinferno_amd.expr.Sym` over user arrays), or a tuple of two for
        the product transform; ``const``: the subtrahend of "sub" / "prod_sub"; ``aux``: the grid of "gridindex".
        Identity of the SOURCE (and aux) arrays keys the engine cache."""
        self.transform, self.source, self.const, self.aux, self._cache, self._expr = transform, source, float(const), aux, None, None

    def key(self):
        """Cache identity without building the expression (a model function makes its columns on every call)."""
        src = self.source
        ids = tuple(a.key if isinstance(a, E.Sym) else id(a) for a in src) if isinstance(src, tuple) else (src.key if isinstance(src, E.Sym) else id(src),)
        return (self.transform, self.const) + ids + ((id(self.aux),) if self.aux is not None else ())

    def expr(self):
        """The column as a setup expression (gwinferno_amd.expr): what the device ingest kernel and :meth:`values` evaluate."""
        if self._expr is None:
            t = self.transform
            if t == "prod_sub":  # a * b - const  (e.g. m2 - mmin = q m1 - mmin)
                a, b = (E.Sym.of(v) for v in self.source)
                e = a * b - self.const
            else:
                x = E.Sym.of(self.source)
                if t == "gridindex":
                    e = E.gridindex(x, self.aux)
                elif t == "sub":
                    e = x - self.const
                elif t == "logdiv":    # log(x / const)
                    e = E.log(x / self.const)
                elif t == "log1mdiv":  # log(1 - x / const)
                    e = E.log(1.0 - x / self.const)
                elif t == "id":
                    e = x
                elif t == "log":
                    e = E.log(x)
                elif t == "neglog":
                    e = -E.log(x)
                elif t == "log1m":
                    e = E.log(1.0 - x)
                elif t == "log1p":
                    e = E.log(1.0 + x)
                elif t == "abs":
                    e = abs(x)
                else:
                    raise ValueError(t)
            self._expr = e
        return self._expr

    def values(self):
        if self._cache is None:
            self._cache = np.ascontiguousarray(np.broadcast_to(self.expr().numpy(), self.expr().shape), dtype=np.float64)
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


class LogValues:
    """A theta-independent per-sample array that is ALREADY a logarithm (``jnp.log(samps["prior"])`` in the
    log-space model functions, examples/config_files/model.py:21-22; analysis.py:401-402).  ``key`` identifies it
    for the engine cache: the identity of the array the logarithm was taken of (:func:`log`; the logarithm itself is
    then only computed when an engine is bound), or a content hash (blake2b over the bytes) for an anonymous array (a model function
    recomputes ``log(prior)`` on every call)."""

    __slots__ = ("_values", "key", "_source", "_expr")

    def __init__(self, values=None, source=None, expr=None):
        self._values, self._source, self._expr = values, source, expr
        if expr is not None:      # a setup expression that IS the logarithm
            self.key = ("log-expr",) + expr.key
        elif source is not None:
            self.key = ("log-of", id(source))
        else:
            import hashlib

            v = self._values = np.ascontiguousarray(values, dtype=np.float64)
            # content hash of the whole array (one pass at memory bandwidth): two different log(prior) arrays can never
            # share a cached engine.  Arrays keyed by identity (`log(source)`, model data) must not be mutated in place.
            self.key = ("log-values", v.shape, hashlib.blake2b(v.tobytes() if v.size < (1 << 16) else memoryview(v).cast("B"), digest_size=16).hexdigest())

    @property
    def values(self):
        if self._values is None:
            if self._expr is not None:
                self._values = np.broadcast_to(np.asarray(self._expr.numpy(), dtype=np.float64), self._expr.shape)
            else:
                with np.errstate(all="ignore"):
                    self._values = np.log(np.asarray(self._source, dtype=np.float64))
        return self._values

    @property
    def shape_source(self):
        """An array with the shape of the values, without computing them."""
        if self._expr is not None:
            return self._expr
        return self._source if self._values is None else self._values


def log(x):
    """``jnp.log`` for per-sample data inside a log-space model function: a :class:`LogValues` remembering
    which array it came from, so that repeated model calls hit the same cached engine and never recompute the
    logarithm."""
    return LogValues(source=x)


def static_key(a):
    """Cache identity of one ``log_static`` entry."""
    return a.key if isinstance(a, (LogValues, E.Sym)) else id(a)


def static_log_expr(a):
    """log of one ``log_static`` entry as a setup expression."""
    if isinstance(a, LogValues):
        if a._expr is not None:
            return a._expr
        return E.log(E.Sym.src(a._source)) if a._values is None else E.Sym.src(a._values)
    return E.log(E.Sym.of(a))


def static_log_values(a):
    """log of one ``log_static`` entry as a float64 array."""
    if isinstance(a, LogValues):
        return a.values
    with np.errstate(all="ignore"):
        return np.log(np.asarray(a, dtype=np.float64))


class Factor:
    """One multiplicative term of a density, bound to ONE side's data."""

    def __init__(self, kind, side, columns, scalars=(), coefs=None, consts=(), n_basis=0, flags=0, mask=None, static_log=None, norm=None,
                 owner=None, norm_owner=None, tag=""):
        # (a model function builds its factors on every evaluation: no per-element conversions here)
        self.kind = kind
        self.side = side
        self.columns = columns if type(columns) is list else list(columns)  # list[Column], order = cols[] of gwi_term
        self.scalars = scalars if type(scalars) is list else list(scalars)  # hyper-parameter values, order = theta[] of gwi_term
        self.coefs = coefs                    # spline coefficient vector or None
        self.consts = tuple(map(float, consts)) if consts else ()  # p[] of gwi_term
        self.n_basis = int(n_basis)
        self.flags = int(flags)
        # Setup expressions (gwinferno_amd.expr) over the caller's arrays, or zero-argument callables returning them:
        # ``mask`` (False -> sample excluded, weight 0) and ``static_log`` (theta-independent per-sample log factor, e.g.
        # log dVc/dz).  Only bind() looks at them, once per engine, and evaluates them on the device where there is one; a
        # model function builds its factors on EVERY call, so even assembling the expression is deferred (the callable).
        # Plain arrays are accepted and become sources of the expression.
        self._mask = mask
        self._static_log = static_log
        self.norm = norm                      # GridNorm dividing this factor, or None
        self.owner = owner                    # model object (pairs PE and injection sides)
        self.norm_owner = norm_owner if norm_owner is not None else owner  # normalisers are shared per owner
        self.tag = tag

    def mask_expr(self):
        """The validity mask as a boolean setup expression, or None."""
        if callable(self._mask):
            self._mask = self._mask()
        if self._mask is not None and not isinstance(self._mask, E.Sym):
            self._mask = E.Sym.src(np.ascontiguousarray(self._mask))
        return self._mask

    def static_log_expr(self):
        if callable(self._static_log):
            self._static_log = self._static_log()
        if self._static_log is not None and not isinstance(self._static_log, E.Sym):
            self._static_log = E.Sym.src(np.ascontiguousarray(self._static_log, dtype=np.float64))
        return self._static_log

    @property
    def mask(self):
        """The mask as a bool array (host evaluation of :meth:`mask_expr`)."""
        e = self.mask_expr()
        return None if e is None else np.broadcast_to(E._as_bool(e.numpy()), e.shape)

    @property
    def static_log(self):
        e = self.static_log_expr()
        return None if e is None else np.broadcast_to(np.asarray(e.numpy(), dtype=np.float64), e.shape)

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
        if _is_number(other):
            with np.errstate(all="ignore"):
                return Density(self.factors, self.side, self.log_static, self.log_const + np.log(float(other)))
        return Density(self.factors, self._merge_side(side_of(other)), self.log_static + [(1.0, other)], self.log_const)

    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, Density):
            raise TypeError("division by a lazy density is not supported")
        if _is_number(other):
            return Density(self.factors, self.side, self.log_static, self.log_const - np.log(float(other)))
        return Density(self.factors, self._merge_side(side_of(other)), self.log_static + [(-1.0, other)], self.log_const)

    def __repr__(self):
        return f"Density(side={self.side}, factors={[f.kind for f in self.factors]})"


class LogDensity(Density):
    """The logarithm of a :class:`Density`: what ``log_prob`` of the distribution classes returns
    (gwinferno_amd.numpyro_distributions).  ``+`` / ``-`` of the reference's log-space model functions
    (analysis.py:401-402; examples/config_files/model.py:21-22) map onto products / quotients of the
    underlying densities; ``sum([...])`` works (``0 + x``)."""

    @staticmethod
    def _of(d):
        return LogDensity(d.factors, d.side, d.log_static, d.log_const)

    def _shift(self, other, sgn):
        if isinstance(other, LogValues) or np.ndim(other) > 0:
            lv = other if isinstance(other, LogValues) else LogValues(other)
            return LogDensity(self.factors, self._merge_side(side_of(lv.shape_source)), self.log_static + [(sgn, lv)], self.log_const)
        return LogDensity(self.factors, self.side, self.log_static, self.log_const + sgn * float(other))

    def __add__(self, other):
        if isinstance(other, Density):
            return LogDensity._of(Density.__mul__(self, other))
        return self._shift(other, 1.0)

    __radd__ = __add__

    def __sub__(self, other):
        if isinstance(other, Density):
            raise TypeError("subtracting a lazy log-density (division by a density) is not supported")
        return self._shift(other, -1.0)


def where_finite(density):
    """Stand-in for ``jnp.where(jnp.isnan(w) | jnp.isinf(w), 0, w)`` (tests/inference_test.py:172, 260):
    the engine always applies that guard, so this is the identity on lazy densities."""
    return density


class LazyNorm:
    """Handle for a normaliser value (e.g. ``z_model.normalization(lamb)``, parametric.py:123-124),
    resolved by the engine to the Z it integrates in the same launch."""

    def __init__(self, owner, scalars=(), coefs=None):
        self.owner, self.scalars, self.coefs = owner, list(scalars), coefs
