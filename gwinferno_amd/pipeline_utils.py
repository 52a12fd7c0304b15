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

The file readers of that module (``load_pe_and_injections_as_dict``: an arviz InferenceData with NetCDF-4 groups)
need arviz / h5py and are not restated; ``gwinferno_amd.catalog`` reads the PE tensor and applies the injection cuts.
"""
import numpy as np

from . import likelihood as L
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


def bspline_mass_prior(m_nsplines=None, q_nsplines=None, m_tau=1, q_tau=1, name=None, m_cs_sig=15, q_cs_sig=5, m_deg=1, q_deg=1):
    """pipeline/utils.py:163-182."""
    name = "_" + name if name is not None else ""
    if m_nsplines is not None:
        mass_cs = _sample_normal("mass_cs" + name, m_cs_sig, m_nsplines)
        _factor("mass_smoothing_prior" + name, apply_difference_prior(mass_cs, m_tau, degree=m_deg))
    if q_nsplines is not None:
        q_cs = _sample_normal("q_cs" + name, q_cs_sig, q_nsplines)
        _factor("q_smoothing_prior" + name, apply_difference_prior(q_cs, q_tau, degree=q_deg))
    if m_nsplines is not None and q_nsplines is None:
        return mass_cs
    if m_nsplines is None and q_nsplines is not None:
        return q_cs
    if m_nsplines is None and q_nsplines is None:
        raise AssertionError("number of mass splines or q splines must be specified.")
    return mass_cs, q_cs


def bspline_spin_prior(a_nsplines=None, ct_nsplines=None, a_tau=None, ct_tau=None, name=None, IID=False, a_cs_sig=5, ct_cs_sig=5, a_deg=2, ct_deg=2):
    """pipeline/utils.py:185-208."""
    name = "_" + name if name is not None else ""
    if IID:
        a_cs = _sample_normal("a_cs" + name, a_cs_sig, a_nsplines)
        _factor("a_smoothing_prior" + name, apply_difference_prior(a_cs, a_tau, degree=a_deg))
        ct_cs = _sample_normal("tilt_cs" + name, ct_cs_sig, ct_nsplines)
        _factor("ct_smoothing_prior" + name, apply_difference_prior(ct_cs, ct_tau, degree=ct_deg))
        return a_cs, ct_cs
    a1_cs = _sample_normal("a1_cs" + name, a_cs_sig, a_nsplines)
    _factor("a1_smoothing_prior" + name, apply_difference_prior(a1_cs, a_tau, degree=a_deg))
    a2_cs = _sample_normal("a2_cs" + name, a_cs_sig, a_nsplines)
    _factor("a2_smoothing_prior" + name, apply_difference_prior(a2_cs, a_tau, degree=a_deg))
    ct1_cs = _sample_normal("tilt1_cs" + name, ct_cs_sig, ct_nsplines)
    _factor("ct1_smoothing_prior" + name, apply_difference_prior(ct1_cs, ct_tau, degree=ct_deg))
    ct2_cs = _sample_normal("tilt2_cs" + name, ct_cs_sig, ct_nsplines)
    _factor("ct2_smoothing_prior" + name, apply_difference_prior(ct2_cs, ct_tau, degree=ct_deg))
    return a1_cs, ct1_cs, a2_cs, ct2_cs


def bspline_redshift_prior(z_nsplines=None, z_tau=None, name=None, z_cs_sig=1, z_deg=2):
    """pipeline/utils.py:211-216: the first coefficient is pinned to 0."""
    name = "_" + name if name is not None else ""
    z_cs = _sample_normal("z_cs" + name, z_cs_sig, z_nsplines - 1)
    npro = L._numpyro()
    if npro is not None:
        import jax.numpy as jnp

        z_cs = jnp.concatenate([jnp.zeros(1), z_cs])
    else:
        z_cs = np.concatenate([np.zeros(1), z_cs])
    _factor("z_smoothing_prior" + name, apply_difference_prior(z_cs, z_tau, degree=z_deg))
    return z_cs


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
