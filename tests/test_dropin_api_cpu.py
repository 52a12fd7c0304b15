"""CPU: argument handling of the drop-in API that needs no device (the reference's error behaviour,
analysis.py:237-243, and the lazy-density algebra)."""
import numpy as np
import pytest

from gwinferno_amd import models as M
from gwinferno_amd.lazy import Density
from gwinferno_amd.likelihood import hierarchical_likelihood
from gwinferno_amd.synthetic import make_catalog


def _weights():
    pe, inj, total = make_catalog(3, 16, 40, seed=2)
    zm = M.PowerlawRedshiftModel(pe["redshift"], inj["redshift"])

    def w(d):
        return M.powerlaw_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], -2.0, 1.0, 5.0, 100.0) * zm(d["redshift"], 2.7) / d["prior"]

    return w(pe), w(inj), zm, total


def test_max_variance_cut_argument_check_matches_reference():
    pw, iw, zm, total = _weights()
    with pytest.raises(ValueError, match="max_variance_cut is True which requires"):
        hierarchical_likelihood(pw, iw, total, 3, 1.0, surveyed_hypervolume=zm.normalization(2.7), max_variance_cut=True)  # min_neff_cut defaults to True


def test_out_of_scope_branches_raise():
    pw, iw, zm, total = _weights()
    with pytest.raises(NotImplementedError):
        hierarchical_likelihood(pw, iw, total, 3, 1.0, surveyed_hypervolume=zm.normalization(2.7), categorical=True)
    with pytest.raises(TypeError):
        hierarchical_likelihood(np.ones((3, 16)), np.ones(40), total, 3, 1.0, surveyed_hypervolume=zm.normalization(2.7))


def test_density_algebra():
    pw, iw, zm, total = _weights()
    assert isinstance(pw, Density) and pw.side == "pe" and iw.side == "inj"
    assert len(pw.factors) == 3 and len(pw.log_static) == 1  # PL q, PL m1, PL z; / prior
    half = 0.5 * pw
    assert half.log_const == pytest.approx(np.log(0.5))
    with pytest.raises(ValueError):
        pw * iw  # PE and injection products cannot be mixed


def test_model_shapes_and_truncation_like_reference_tests():
    """tests/models/bsplines/separable_test.py:93-97 and parametric_test.py: masks mark samples
    outside [mmin, mmax] / z > zmax as zero-density."""
    pe, inj, _ = make_catalog(3, 16, 40, seed=2)
    m = M.BSplineMass(10, pe["mass_1"], inj["mass_1"], mmin=5.0, mmax=100.0)
    f = m(np.zeros(10), pe_samples=True).factors[0]
    assert f.mask.shape == pe["mass_1"].shape
    assert np.array_equal(f.mask, (pe["mass_1"] >= 5.0) & (pe["mass_1"] <= 100.0))
    zm = M.PowerlawRedshiftModel(pe["redshift"], inj["redshift"])
    fz = zm(pe["redshift"], 2.0).factors[0]
    assert np.array_equal(fz.mask, pe["redshift"] <= zm.zmax)
    with pytest.raises(ValueError):
        zm(pe["redshift"][:, :3], 2.0)  # not the array the model was built with


def test_smoothing_prior_matches_reference_golden():
    import os

    from golden_util import GOLDEN_DIR

    from gwinferno_amd.smoothing import apply_difference_prior

    z = np.load(os.path.join(GOLDEN_DIR, "terms.npz"))
    got = [apply_difference_prior(z["smoothing/coefs"], tau, deg) for tau, deg in ((1.0, 1), (25.0, 2), (5.0, 3))]
    assert np.allclose(got, z["smoothing/values"], rtol=1e-14)
    assert apply_difference_prior(np.ones(10), 5) == 0  # tests/models/bsplines/smoothing_test.py:18-21


def test_numpyro_distribution_log_prob_faces_bind_like_models():
    """Powerlaw / PowerlawRedshift .log_prob (numpyro_distributions.py:127-136, 186-195) produce the same
    bound terms as the corresponding model functions."""
    from bound_eval import log_weights

    from gwinferno_amd.engine import bind
    from gwinferno_amd.numpyro_distributions import Powerlaw, PowerlawRedshift
    from oracle import numpy_oracle as O

    pe, inj, total = make_catalog(3, 16, 40, seed=2)
    zgrid = np.linspace(1e-9, 1.9, 1000)
    dV = O.planck15_lvk().dVc_dz(zgrid)

    pm, pz = Powerlaw(-2.3, 5.0, 100.0), PowerlawRedshift(2.7, 1.9, zgrid, dV)  # one object per model call (analysis.py:381-399)

    def w(d):
        return pm.log_prob(d["mass_1"]) * pz.log_prob(d["redshift"]) / d["prior"]

    wp, wi = w(pe), w(inj)
    bm = bind(wp, wi)
    lpe, linj, norms = log_weights(bm, bm.theta_of(wp))
    # direct evaluation of the reference formulas
    with np.errstate(all="ignore"):
        for d, got in ((pe, lpe), (inj, linj)):
            lp_m = -2.3 * np.log(d["mass_1"]) + np.log((1 - 2.3) / (100.0 ** (1 - 2.3) - 5.0 ** (1 - 2.3)))
            lp_m = np.where((d["mass_1"] < 5.0) | (d["mass_1"] > 100.0), -np.inf, lp_m)
            norm = np.trapezoid(dV * (1 + zgrid) ** 1.7, zgrid)
            lp_z = np.log(np.interp(d["redshift"], zgrid, dV)) + 1.7 * np.log(1 + d["redshift"]) - np.log(norm)
            ref = lp_m + lp_z - np.log(d["prior"])
            ok = np.isfinite(ref)
            assert np.array_equal(np.isfinite(got), ok)
            assert np.max(np.abs(got[ok] - ref[ok])) < 1e-11


def test_prior_gradient_matches_finite_differences():
    from gwinferno_amd.sampling import GaussianSmoothingPrior

    rng = np.random.default_rng(0)
    th = rng.normal(size=20)
    prior = GaussianSmoothingPrior(20).normal(slice(0, 20), 3.0).smoothing(slice(2, 14), 5.0, 1).smoothing(slice(4, 20), 25.0, 2)
    lp, g = prior(th)
    for i in range(20):
        e = np.zeros(20)
        e[i] = 1e-6
        fd = (prior(th + e)[0] - prior(th - e)[0]) / 2e-6
        assert abs(fd - g[i]) < 1e-6 * max(1.0, abs(g[i]))


def test_bijector_roundtrip_and_jacobian():
    from gwinferno_amd.sampling import Bijector

    b = Bijector(4).interval(1, 5.0, 100.0).positive(2).interval(3, 0.0, 1.0)
    th = np.array([-2.5, 35.0, 4.0, 0.1])
    u = b.inverse(th)
    th2, dth, dlogj, logj = b.forward(u)
    assert np.allclose(th2, th)
    for i in range(4):
        e = np.zeros(4)
        e[i] = 1e-6
        fd = (b.forward(u + e)[0][i] - b.forward(u - e)[0][i]) / 2e-6
        assert abs(fd - dth[i]) < 1e-6
        fdj = (b.forward(u + e)[3] - b.forward(u - e)[3]) / 2e-6
        assert abs(fdj - dlogj[i]) < 1e-6


def test_lazy_products_do_no_array_work_until_bound():
    """A model function runs on every evaluation: building its lazy products must not touch the catalog (masks, static
    logs and log(prior) stay unevaluated until an engine is bound), engines are found again by the identity of the
    caller's arrays -- also when those are float32 or ndarray subclasses (memmaps) that have to be converted -- and the
    log-space algebra maps onto the same products."""
    from gwinferno_amd import lazy
    from gwinferno_amd import models as M
    from gwinferno_amd import numpyro_distributions as D
    from gwinferno_amd.engine import bind, structure_key

    class Sub(np.ndarray):  # stands for np.memmap and friends
        pass

    pe, inj, total = make_catalog(3, 16, 40, seed=2)
    pe = {k: np.ascontiguousarray(v, dtype=np.float64).view(Sub) for k, v in pe.items()}
    inj = {k: np.asarray(v, dtype=np.float32) if k == "mass_ratio" else v for k, v in inj.items()}  # one float32 column
    z_model = M.PowerlawRedshiftModel(z_pe=pe["redshift"], z_inj=inj["redshift"])

    def linear(d):
        return M.plpeak_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], -2.3, 1.1, 5.0, 100.0, 33.0, 4.0, 0.1) * z_model(d["redshift"], 2.5) / d["prior"]

    def log_space(d):
        return D.Powerlaw(-2.3, 5.0, 100.0).log_prob(d["mass_1"]) + D.Powerlaw(1.1, 0.02, 1.0).log_prob(d["mass_ratio"]) - lazy.log(d["prior"])

    for build in (linear, log_space):
        wp, wi = build(pe), build(inj)
        for w in (wp, wi):
            from gwinferno_amd.expr import Sym

            assert all(f._mask is None or callable(f._mask) or isinstance(f._mask, Sym) for f in w.factors)            # no comparison has run
            assert all(f._static_log is None or callable(f._static_log) or isinstance(f._static_log, Sym) for f in w.factors)   # ... no logarithm / interpolation
            assert all(c._cache is None for f in w.factors for c in f.columns)             # no transform has run
            assert all(a._values is None for _, a in w.log_static if isinstance(a, lazy.LogValues))
        assert structure_key(wp, wi) == structure_key(build(pe), build(inj))  # stable across calls: ONE cached engine serves them all
        bm = bind(wp, wi)  # ... and only now masks, transforms and logs are computed
        assert bm.n_theta > 0 and all(c.dtype == np.float64 and type(c) is np.ndarray for c in bm.pe_cols + bm.inj_cols)
    # log-space algebra: sum([...]) and scalar shifts
    a = D.Powerlaw(-2.0, 5.0, 100.0).log_prob(np.asarray(inj["mass_1"]))
    total_ld = sum([a, D.Powerlaw(1.0, 0.02, 1.0).log_prob(np.asarray(inj["mass_ratio"]))]) + 0.5
    assert isinstance(total_ld, lazy.LogDensity) and len(total_ld.factors) == 2 and abs(total_ld.log_const - 0.5) < 1e-15
    with pytest.raises(TypeError):
        a - a
