"""``log_prob`` faces of the reference's NumPyro distributions that sit on the likelihood path
(gwinferno/numpyro_distributions.py): what ``construct_hierarchical_model`` sums
(pipeline/analysis.py:401-402).  ``log_prob(value)`` returns a lazy density for ``value``'s sample
set, composable with ``*`` like every other model in this package (log-space sums of the reference
== products here).  Sampling / cdf / icdf are not on the log-prob path and are not provided."""
import numpy as np

from . import _native as N
from .interpolation import trapezoid_weights
from .lazy import Column, Density, Factor, GridNorm, LazyNorm, side_of


class Powerlaw:
    """numpyro_distributions.py:101-153: x^alpha on [minimum, maximum]."""

    def __init__(self, alpha, minimum=0.0, maximum=1.0, low=0.0, high=1.0, validate_args=None):
        self.alpha, self.minimum, self.maximum = alpha, float(minimum), float(maximum)

    def log_prob(self, value):
        value = np.asarray(value, dtype=np.float64)
        side = side_of(value)
        with np.errstate(all="ignore"):
            mask = ~((value < self.minimum) | (value > self.maximum))  # :131-136
        return Density([Factor(N.TERM_POWERLAW, side, [Column("log", value)], [self.alpha], consts=(self.minimum, self.maximum), mask=mask)], side)


class PowerlawRedshift:
    """numpyro_distributions.py:156-201: dVc/dz (1+z)^(lamb-1) / trapz(...) on a caller-supplied grid;
    dVc/dz at the samples by linear interpolation into that grid (:189-190)."""

    def __init__(self, lamb, maximum, zgrid, dVcdz, low=0.0, high=1000.0, validate_args=None):
        self.lamb, self.maximum = lamb, float(maximum)
        self.zs = np.asarray(zgrid, dtype=np.float64)
        self.dVdc_ = np.asarray(dVcdz, dtype=np.float64)
        with np.errstate(all="ignore"):
            self._norm = GridNorm(trapezoid_weights(self.zs), lb=np.log(self.dVdc_), l1=np.log(1.0 + self.zs), expo_add=-1.0)

    @property
    def norm(self):
        """Lazy handle for the normaliser (``surveyed_hypervolume=pop_models["redshift"].norm``, analysis.py:410)."""
        return LazyNorm(self, [self.lamb])

    def log_prob(self, value, dVdc=None):
        value = np.asarray(value, dtype=np.float64)
        side = side_of(value)
        if dVdc is None:
            dVdc = np.interp(value, self.zs, self.dVdc_)
        with np.errstate(all="ignore"):
            mask = value <= self.maximum
            f = Factor(N.TERM_POWERLAW_REDSHIFT, side, [Column("log1p", value)], [self.lamb], mask=mask, static_log=np.log(dVdc), owner=self, tag="plz")
        f.norm = GridNorm(self._norm.tw, lb=self._norm.lb, l1=self._norm.l1, expo_param=(f, 0), expo_add=-1.0)
        return Density([f], side)
