"""CPU: pin the NumPy oracle (oracle/numpy_oracle.py) to the golden vectors produced by the
unmodified reference (tests/golden/make_golden.py).  Tolerances: 1e-12 relative for per-sample
densities and design matrices, 1e-11 for reduced sites (different summation order only)."""
import json
import os

import numpy as np
import pytest
from golden_util import CASES, GOLDEN_DIR, GoldenCase, rel_err

from oracle import numpy_oracle as O


def _load(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


def test_design_matrices_and_norms():
    z = _load("bases.npz")
    meta = json.loads(str(z["meta"]))
    kind = {"BSpline": "B", "LogXBSpline": "logX", "LogYBSpline": "logY", "LogXLogYBSpline": "logXlogY"}
    for i, m in enumerate(meta):
        basis = O.SplineBasis(kind[m["cls"]], m["n"], tuple(m["xrange"]), normalize=True)
        assert np.allclose(basis.knots, z[f"{i}/knots"], rtol=0, atol=1e-14)
        dm = basis.design(z[f"{i}/xs"])
        ref = z[f"{i}/design"]
        assert dm.shape == ref.shape
        assert np.array_equal(np.isneginf(dm), np.isneginf(ref)), m
        fin = np.isfinite(ref)
        assert np.max(np.abs(dm[fin] - ref[fin])) < 1e-14, m
        assert np.allclose(basis.grid, z[f"{i}/grid"], rtol=0, atol=0)
        assert rel_err(basis.norm(z[f"{i}/coefs"]), z[f"{i}/norm"]) < 1e-13, m
        with np.errstate(all="ignore"):
            assert rel_err(basis.project(dm, z[f"{i}/coefs"]), z[f"{i}/project"]) < 1e-12, m


def test_design_matrix_matches_scipy():
    """Same check the reference makes (tests/interpolation_test.py:50-55)."""
    from scipy.interpolate import BSpline as SciBSpline

    basis = O.SplineBasis("B", 10, (0.0, 1.0), normalize=False)
    gr = np.linspace(0, 1, 1000)
    ours = basis.design(gr).T
    theirs = SciBSpline(basis.knots, np.eye(10), 3)(gr)
    # scipy's last interval is closed on the right as well; the reference (and we) return the
    # (1/6, 2/3, 1/6) taps there, identical to scipy's value at x = 1
    assert np.allclose(ours, theirs)


def test_spline_flavours_integrate_to_one():
    """tests/interpolation_test.py:57-85: every normalised flavour integrates to 1 (3 places)."""
    rng = np.random.default_rng(5)
    gr, grid = np.linspace(0, 1, 1000), np.linspace(0.001, 1, 1000)
    for kind, xr, g, cs in (
        ("B", (0, 1), gr, rng.uniform(size=10)),
        ("logY", (0, 1), gr, rng.normal(size=10)),
        ("logX", (0.001, 1), grid, rng.uniform(size=10)),
        ("logXlogY", (0.001, 1), grid, rng.normal(size=10)),
    ):
        b = O.SplineBasis(kind, 10, xr, normalize=True)
        assert abs(np.trapezoid(b.project(b.design(g), cs), g) - 1.0) < 5e-4, kind


def test_term_densities():
    z = _load("terms.npz")
    m1, q = z["m1"], z["q"]
    for tag, a in zip(("a", "b", "neg1", "zero"), z["powerlaw_alphas"]):
        assert rel_err(O.powerlaw_pdf(m1, a, 5.0, 100.0), z[f"powerlaw_pdf/{tag}"]) < 1e-13
        got, ref = O.powerlaw_pdf(q, a, 5.0 / m1, 1.0), z[f"powerlaw_q/{tag}"]
        ok = np.isfinite(ref)
        assert np.array_equal(np.isfinite(got), ok)
        assert rel_err(got[ok], ref[ok]) < 1e-13
    mu, sig, lo, hi = z["truncnorm_params"]
    assert rel_err(O.truncnorm_pdf(m1, mu, sig, lo, hi), z["truncnorm_pdf"]) < 1e-13
    al, lo, hi, mpp, sigpp, lam = z["plpeak_params"]
    assert rel_err(O.plpeak_primary_pdf(m1, al, lo, hi, mpp, sigpp, lam), z["plpeak_primary_pdf"]) < 1e-13
    got = O.plpeak_primary_ratio_pdf(m1, q, al, float(z["plpeak_ratio_beta"]), lo, hi, mpp, sigpp, lam)
    ref = z["plpeak_primary_ratio_pdf"]
    ok = np.isfinite(ref)
    assert rel_err(got[ok], ref[ok]) < 1e-13
    a, b = z["beta_params"]
    got, ref = O.betadist(z["a"], a, b), z["betadist"]
    ok = np.isfinite(ref)
    assert np.array_equal(np.isfinite(got), ok) and rel_err(got[ok], ref[ok]) < 1e-13
    xi, sg = z["tilt_params"]
    assert rel_err(O.mixture_isoalign_spin_tilt(z["ct"], xi, sg), z["mixture_isoalign_spin_tilt"]) < 1e-13


def test_densities_vs_scipy_stats():
    """The reference's own analytic pins (tests/distributions_test.py:30-88, rtol 1e-5)."""
    from scipy.stats import beta, truncnorm, truncpareto

    x = np.linspace(2, 55, 1000)
    np.testing.assert_allclose(O.powerlaw_pdf(x, -3.2, 3.0, 50.0), truncpareto.pdf(x, 2.2, 50.0 / 3.0, loc=0.0, scale=3.0), rtol=1e-5)
    x = np.linspace(-1, 1.2, 50)
    np.testing.assert_allclose(O.truncnorm_pdf(x, 0.3, 1.4, -0.8, 1.0), truncnorm.pdf(x, (-0.8 - 0.3) / 1.4, (1.0 - 0.3) / 1.4, loc=0.3, scale=1.4), rtol=1e-5)
    x = np.linspace(0, 1, 50)
    np.testing.assert_allclose(O.betadist(x, 2, 3), beta.pdf(x, 2, 3), rtol=1e-5)


def test_redshift_model_and_cosmology():
    z = _load("terms.npz")
    zm = O.PowerlawRedshift(z["z_pe"], z["z_inj"])
    assert np.array_equal([zm.zmin, zm.zmax], z["z_model/zmin_zmax"])
    assert rel_err(zm.dVdz_by_rank[2], z["z_model/dVdz_pe"]) < 1e-14
    for i, lamb in enumerate(z["z_lamb"]):
        assert rel_err(zm.normalization(lamb), z["z_model/norm"][i]) < 1e-13
        assert rel_err(zm(z["z_pe"], lamb), z["z_model/pe"][i]) < 1e-13
        assert rel_err(zm(z["z_inj"], lamb), z["z_model/inj"][i]) < 1e-13
    # product-side cosmology table == oracle's literal recurrence
    from gwinferno_amd.cosmology import planck15_lvk

    zz = np.linspace(0, 3, 777)
    assert rel_err(planck15_lvk().dVc_dz(zz)[1:], O.planck15_lvk().dVc_dz(zz)[1:]) < 1e-14


def test_smoothing_prior():
    z = _load("terms.npz")
    got = [O.apply_difference_prior(z["smoothing/coefs"], tau, deg) for tau, deg in ((1.0, 1), (25.0, 2), (5.0, 3))]
    assert rel_err(got, z["smoothing/values"]) < 1e-14
    assert O.apply_difference_prior(np.ones(10), 5) == 0  # tests/models/bsplines/smoothing_test.py:18-21


@pytest.mark.parametrize("name", CASES)
def test_full_likelihood_sites(name):
    case = GoldenCase(name)
    comp = O.COMPOSITIONS[case.composition](case.pe, case.inj, mmin=case.meta["mmin"], mmax=case.meta["mmax"])
    w_pe = comp.weights(case.point(0), True)
    w_inj = comp.weights(case.point(0), False)
    assert np.array_equal(w_pe == 0, case.weights_pe == 0)
    assert np.array_equal(w_inj == 0, case.weights_inj == 0)
    assert rel_err(w_pe, case.weights_pe) < 1e-12
    assert rel_err(w_inj, case.weights_inj) < 1e-12
    for fs, flags in case.flagsets.items():
        for i in range(case.n_points):
            got = comp.evaluate(case.point(i), case.total_inj, tobs=case.tobs, **flags)
            for site, ref in case.sites[fs].items():
                if site == "rate_return":
                    site_got = got["rate"]
                else:
                    site_got = got[site]
                assert rel_err(site_got, ref[i]) < 1e-11, (name, fs, i, site, site_got, ref[i])


@pytest.mark.parametrize("name", ["pl_test", "bspline_test"])
def test_fd_gradient_reproduces(name):
    case = GoldenCase(name)
    comp = O.COMPOSITIONS[case.composition](case.pe, case.inj, mmin=case.meta["mmin"], mmax=case.meta["mmax"])
    g = O.fd_gradient(comp, case.point(0), case.total_inj, log=False, min_neff_cut=False)
    for k, ref in case.fdgrad[0].items():
        assert np.allclose(g[k], ref, rtol=1e-7, atol=1e-8), k


def test_numpyro_distribution_log_probs():
    """Powerlaw / PowerlawRedshift / BSplineDistribution .log_prob (numpyro_distributions.py:127-136, 186-195,
    296-301) against the reference's own outputs, boundary values and the alpha = -1 branch included."""
    z = _load("terms.npz")
    x = z["m1"]
    for tag, a in zip(("a", "b", "neg1", "zero"), z["powerlaw_alphas"]):
        assert rel_err(O.powerlaw_log_prob(x, a, 5.0, 100.0), z[f"dist/powerlaw/{tag}"]) < 1e-12
    for i, lamb in enumerate(z["z_lamb"]):
        d = O.PowerlawRedshiftDistribution(lamb, float(z["dist/powerlaw_redshift/maximum"]), z["dist/z_grid"], z["dist/z_dVcdz"])
        assert rel_err(d.norm, z["dist/powerlaw_redshift/norm"][i]) < 1e-13
        assert rel_err(d.log_prob(z["z_inj"]), z["dist/powerlaw_redshift/inj"][i]) < 1e-11
    gr, grx = np.linspace(0, 1, 1000), np.linspace(0.001, 1, 1000)
    for tag, kind, xr, g in (("bspline", "B", (0, 1), gr), ("logy", "logY", (0, 1), gr), ("logx", "logX", (0.01, 1), grx), ("logxy", "logXlogY", (0.001, 1), grx)):
        d = O.BSplineDistribution(z["dist/bspline/cs"], g, O.SplineBasis(kind, 20, xr, True).design(g))
        assert rel_err(d.norm, z[f"dist/bspline/{tag}_norm"]) < 1e-13
        assert np.max(np.abs(d.log_prob(z["dist/bspline/value"]) - z[f"dist/bspline/{tag}"])) < 1e-12
