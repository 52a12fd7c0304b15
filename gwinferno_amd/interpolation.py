"""Host-side descriptors of the reference's 1-D cubic B-spline bases on uniform knots
(gwinferno/interpolation.py:236-449).  The reference evaluates the Cox-de Boor recursion into a
dense ``(N_basis, N_samples)`` design matrix at construction (:128-175) and contracts it with the
coefficients every step (:304, :393); this package never materialises that matrix -- the engine
recomputes the 4 non-zero taps per sample from the knot coordinate.  What is left on the host is
bookkeeping: the coordinate transform, the closed domain, and the normalisation grid with its
trapezoid weights (:280-291; grids :343, :378, :433).
"""
import numpy as np


def trapezoid_weights(x):
    """Weights w with ``sum(w * y) == trapz(y, x)`` (up to rounding)."""
    x = np.asarray(x, dtype=np.float64)
    w = np.empty_like(x)
    d = np.diff(x)
    w[0] = 0.5 * d[0]
    w[-1] = 0.5 * d[-1]
    w[1:-1] = 0.5 * (d[1:] + d[:-1])
    return w


class DesignMatrix(np.ndarray):
    """``basis.bases(xs)`` as a dense array that remembers which basis and which points produced it, so that
    consumers which the reference hands a design matrix (BSplineDistribution's ``grid_dmat``,
    numpyro_distributions.py:266-279) can recompute the taps on the device instead of reading the matrix."""

    basis = None
    xs = None

    def __array_finalize__(self, obj):
        if obj is not None and obj.shape == self.shape:  # views / copies keep the tag; slices and reductions drop it
            self.basis, self.xs = getattr(obj, "basis", None), getattr(obj, "xs", None)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):  # arithmetic yields plain arrays: no longer THE design matrix
        plain = tuple(np.asarray(i) if isinstance(i, DesignMatrix) else i for i in inputs)
        if "out" in kwargs:
            kwargs["out"] = tuple(np.asarray(o) if isinstance(o, DesignMatrix) else o for o in kwargs["out"])
        return getattr(ufunc, method)(*plain, **kwargs)


class _Basis:
    """Common descriptor: ``name``, log-X / log-Y flags, number of grid points."""

    log_x = False
    log_y = False
    grid_points = 1000
    name = "BSpline"

    def __init__(self, n_df, xrange=(0, 1), k=4, normalize=False, **_ignored):
        if k != 4:
            raise NotImplementedError("only cubic splines (k=4) are implemented, as used by every reference model")
        if n_df < 4:
            raise ValueError("a cubic B-spline needs at least 4 basis functions")
        self.N = int(n_df)
        self.order = 4
        self.normalize = bool(normalize)
        # log-X bases live on log(xrange) (interpolation.py:337, :427)
        self.xrange = tuple(np.log(np.asarray(xrange, dtype=np.float64))) if self.log_x else (float(xrange[0]), float(xrange[1]))
        self.lo, self.hi = float(self.xrange[0]), float(self.xrange[1])
        # uniform knot vector (interpolation.py:98-106), kept for inspection / tests
        interior = np.linspace(self.lo, self.hi, self.N - self.order + 2)
        self.dx = interior[1] - interior[0]
        self.knots = np.linspace(self.lo - self.dx * 3, self.hi + self.dx * 3, len(interior) + 6)
        self.grid = None
        if self.normalize:
            ends = np.exp(np.asarray(self.xrange)) if self.log_x else np.asarray(self.xrange)
            self.grid = np.linspace(ends[0], ends[1], self.grid_points)

    def coordinate(self, xs):
        with np.errstate(all="ignore"):
            return np.log(xs) if self.log_x else np.asarray(xs, dtype=np.float64)

    def outside(self, coord):
        """Same predicate as the reference (interpolation.py:175, :407, :449)."""
        with np.errstate(all="ignore"):
            return (coord < self.lo) | (coord > self.hi)

    def grid_tables(self):
        """(tw, us) for the normaliser: trapezoid weights in x and spline coordinates of the grid;
        weights are zeroed where a log-Y basis is -inf (exp -> 0)."""
        tw = trapezoid_weights(self.grid)
        us = self.coordinate(self.grid)
        if self.log_y:
            tw = np.where(self.outside(us), 0.0, tw)
        return tw, np.where(np.isfinite(us), us, self.lo)

    # dense evaluation for tests / post-processing only (closed-form taps, NOT the hot path)
    def bases(self, xs):
        coord = self.coordinate(np.asarray(xs, dtype=np.float64))
        n_int = self.N - 3
        u = (coord - self.lo) * (n_int / (self.hi - self.lo))
        k = np.clip(np.floor(u), 0, n_int - 1).astype(np.int64)
        t = u - k
        taps = np.stack([(1 - t) ** 3 / 6, (3 * t**3 - 6 * t**2 + 4) / 6, (-3 * t**3 + 3 * t**2 + 3 * t + 1) / 6, t**3 / 6])
        out = np.zeros((self.N,) + coord.shape)
        flat_idx = np.arange(coord.size).reshape(coord.shape)
        for j in range(4):
            np.add.at(out.reshape(self.N, -1), ((k + j).ravel(), flat_idx.ravel()), taps[j].ravel())
        bad = self.outside(coord)
        out = np.where(bad, -np.inf if self.log_y else 0.0, out).view(DesignMatrix)
        out.basis, out.xs = self, np.asarray(xs, dtype=np.float64)
        return out


class BSpline(_Basis):
    name = "BSpline"


class LogXBSpline(_Basis):
    name = "LogXBSpline"
    log_x = True

    def __init__(self, n_df, xrange=(0.01, 1), normalize=True, **kw):
        super().__init__(n_df, xrange=xrange, normalize=normalize, **kw)


class LogYBSpline(_Basis):
    name = "LogYBSpline"
    log_y = True

    def __init__(self, n_df, xrange=(0, 1), normalize=True, **kw):
        super().__init__(n_df, xrange=xrange, normalize=normalize, **kw)


class LogXLogYBSpline(_Basis):
    name = "LogXLogYBSpline"
    log_x = True
    log_y = True
    grid_points = 1500

    def __init__(self, n_df, xrange=(0.1, 1), normalize=True, **kw):
        super().__init__(n_df, xrange=xrange, normalize=normalize, **kw)
