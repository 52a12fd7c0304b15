"""The model factories and prior helpers of ``gwinferno/pipeline/utils.py`` (:104-216) -- same names, arguments and
return values -- over this package's models, so that a model function written for the reference
(``examples/simple_bspline_example.py:26-94``) reads the same here.

* ``setup_bspline_mass_models`` / ``setup_bspline_spin_models`` / ``setup_powerlaw_spline_redshift_model``
  (:104-160): the BASELINE config-5 factors.
* ``bspline_mass_prior`` / ``bspline_spin_prior`` / ``bspline_redshift_prior`` (:163-216): Normal priors on the
  coefficients plus the P-spline difference penalties as ``numpyro.factor`` sites.  With NumPyro installed they are
  the reference's functions verbatim; without it the coefficients are read from ``likelihood.SAMPLE_VALUES[site]``
  and the factor values are recorded in :data:`PRIOR_FACTORS` (there is no tracer to hand them to).
* :func:`bspline_example_prior` describes the same priors to the library's own sampler
  (``sampling.nuts_engine``): a ``GaussianSmoothingPrior`` + ``Bijector`` over the flat theta of a composition.

``load_pe_and_injections_as_dict`` (:51-96: the arviz InferenceData file with NetCDF-4 groups ``pe_data`` / ``inj_data``) is
here under its own name too, read through the HDF5 C library (``gwinferno_amd.catalog``, ``gwinferno_amd._hdf5``); the same
module reads the NetCDF-3 PE tensor and applies the injection cuts.
"""
import numpy as np

from . import likelihood as L
from .catalog import load_pe_and_injections_as_dict  # noqa: F401  (pipeline/utils.py:51-96)
from .interpolation import LogXLogYBSpline, LogYBSpline
from .models import (BSplineIIDSpinMagnitudes, BSplineIIDSpinTilts, BSplineIndependentSpinMagnitudes, BSplineIndependentSpinTilts, BSplinePrimaryBSplineRatio,
                     PowerlawSplineRedshiftModel)
from .smoothing import apply_difference_prior

PRIOR_FACTORS = {}  # numpyro.factor sites of the prior helpers when numpyro is absent


def setup_bspline_mass_models(pedict, injdict, m_nsplines, q_nsplines, mmin, mmax):
    """pipeline/utils.py:104-118."""
    return BSplinePrimaryBSplineRatio(m_nsplines, q_nsplines, pedict["mass_1"], injdict["mass_1"], pedict["mass_ratio"], injdict["mass_ratio"], m1min=mmin, m2min=mmin, mmax=mmax,
                                      kwargs_m={"basis": LogXLogYBSpline}, kwargs_q={"basis": LogYBSpline})


def setup_bspline_spin_models(pedict, injdict, a1_nsplines, ct1_nsplines, IID=False, a2_nsplines=None, ct2_nsplines=None):
    """pipeline/utils.py:121-146: (mag_model, tilt_model)."""
    if IID:
        tilt_model = BSplineIIDSpinTilts(ct1_nsplines, pedict["cos_tilt_1"], pedict["cos_tilt_2"], injdict["cos_tilt_1"], injdict["cos_tilt_2"], normalize=True)
        mag_model = BSplineIIDSpinMagnitudes(a1_nsplines, pedict["a_1"], pedict["a_2"], injdict["a_1"], injdict["a_2"], normalize=True)
    else:
        tilt_model = BSplineIndependentSpinTilts(ct1_nsplines, ct2_nsplines, pedict["cos_tilt_1"], pedict["cos_tilt_2"], injdict["cos_tilt_1"], injdict["cos_tilt_2"], normalize=True)
        mag_model = BSplineIndependentSpinMagnitudes(a1_nsplines, a2_nsplines, pedict["a_1"], pedict["a_2"], injdict["a_1"], injdict["a_2"], normalize=True)
    return mag_model, tilt_model


def setup_powerlaw_spline_redshift_model(pedict, injdict, z_nsplines):
    """pipeline/utils.py:149-155."""
    return PowerlawSplineRedshiftModel(z_nsplines, pedict["redshift"], injdict["redshift"])


def _sample_normal(site, sigma, n):
    npro = L._numpyro()
    if npro is not None:
        import numpyro.distributions as dist

        return npro.sample(site, dist.Normal(0, sigma), sample_shape=(n,))
    v = np.asarray(L.SAMPLE_VALUES[site], dtype=np.float64)
    if v.shape != (n,):
        raise ValueError(f"SAMPLE_VALUES[{site!r}] must have shape ({n},)")
    return v


def _factor(site, value):
    npro = L._numpyro()
    if npro is not None:
        npro.factor(site, value)
    else:
        PRIOR_FACTORS[site] = float(value)


def _coefficient_block(site, n, sigma, tau, degree, factor_site, prepend_zero=False):
    """One coefficient vector ~ Normal(0, sigma)^n with its difference penalty registered as a factor site."""
    cs = _sample_normal(site, sigma, n)
    if prepend_zero:  # the first redshift coefficient is not sampled (pipeline/utils.py:213-214)
        if L._numpyro() is not None:
            import jax.numpy as jnp

            cs = jnp.concatenate([jnp.zeros(1), cs])
        else:
            cs = np.concatenate([np.zeros(1), cs])
    _factor(factor_site, apply_difference_prior(cs, tau, degree=degree))
    return cs


def _suffix(name):
    return "" if name is None else "_" + name


def bspline_mass_prior(m_nsplines=None, q_nsplines=None, m_tau=1, q_tau=1, name=None, m_cs_sig=15, q_cs_sig=5, m_deg=1, q_deg=1):
    """pipeline/utils.py:163-182: ``mass_cs`` and/or ``q_cs`` (sites ``mass_cs``, ``mass_smoothing_prior``, ``q_cs``,
    ``q_smoothing_prior`` [+ ``_name``])."""
    if m_nsplines is None and q_nsplines is None:
        raise AssertionError("number of mass splines or q splines must be specified.")
    sfx, out = _suffix(name), []
    if m_nsplines is not None:
        out.append(_coefficient_block("mass_cs" + sfx, m_nsplines, m_cs_sig, m_tau, m_deg, "mass_smoothing_prior" + sfx))
    if q_nsplines is not None:
        out.append(_coefficient_block("q_cs" + sfx, q_nsplines, q_cs_sig, q_tau, q_deg, "q_smoothing_prior" + sfx))
    return out[0] if len(out) == 1 else tuple(out)


def bspline_spin_prior(a_nsplines=None, ct_nsplines=None, a_tau=None, ct_tau=None, name=None, IID=False, a_cs_sig=5, ct_cs_sig=5, a_deg=2, ct_deg=2):
    """pipeline/utils.py:185-208: ``(a_cs, tilt_cs)`` when IID, else ``(a1_cs, tilt1_cs, a2_cs, tilt2_cs)``; the sample
    sites are visited in the reference's order (magnitudes first), which is what a seeded NumPyro run depends on."""
    sfx = _suffix(name)
    mags = [("a", "a")] if IID else [("a1", "a1"), ("a2", "a2")]
    tilts = [("tilt", "ct")] if IID else [("tilt1", "ct1"), ("tilt2", "ct2")]
    a = [_coefficient_block(f"{site}_cs{sfx}", a_nsplines, a_cs_sig, a_tau, a_deg, f"{fac}_smoothing_prior{sfx}") for site, fac in mags]
    t = [_coefficient_block(f"{site}_cs{sfx}", ct_nsplines, ct_cs_sig, ct_tau, ct_deg, f"{fac}_smoothing_prior{sfx}") for site, fac in tilts]
    return (a[0], t[0]) if IID else (a[0], t[0], a[1], t[1])


def bspline_redshift_prior(z_nsplines=None, z_tau=None, name=None, z_cs_sig=1, z_deg=2):
    """pipeline/utils.py:211-216: ``z_nsplines - 1`` sampled coefficients behind a leading 0."""
    sfx = _suffix(name)
    return _coefficient_block("z_cs" + sfx, z_nsplines - 1, z_cs_sig, z_tau, z_deg, "z_smoothing_prior" + sfx, prepend_zero=True)


def bspline_example_prior(slices, m_tau=1, q_tau=1, a_tau=25, ct_tau=25, z_tau=1, lamb_sigma=3.0):
    """The priors of examples/simple_bspline_example.py:47-56 for the library's sampler: ``slices`` maps
    ``m1, q, a1, a2, tilt1, tilt2, redshift, lamb`` to their slices of the flat theta (``redshift`` covers all
    ``z_nsplines`` coefficients; the first is pinned to 0).  Returns ``(GaussianSmoothingPrior, Bijector)``."""
    from .sampling import Bijector, GaussianSmoothingPrior

    n = max(sl.stop for sl in slices.values())
    prior, bij = GaussianSmoothingPrior(n), Bijector(n)
    prior.normal(slices["m1"], 15.0).smoothing(slices["m1"], m_tau, 1)
    prior.normal(slices["q"], 5.0).smoothing(slices["q"], q_tau, 1)
    for key, tau in (("a1", a_tau), ("a2", a_tau), ("tilt1", ct_tau), ("tilt2", ct_tau)):
        prior.normal(slices[key], 5.0).smoothing(slices[key], tau, 2)
    z = slices["redshift"]
    prior.normal(slice(z.start + 1, z.stop), 1.0).smoothing(z, z_tau, 2)
    bij.fixed(z.start, 0.0)
    prior.normal(slices["lamb"], lamb_sigma)
    return prior, bij
