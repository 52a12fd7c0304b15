"""GPU (-m gpu): term-level parity.  Each closed-form density of the reference
(gwinferno/distributions.py, models/parametric/parametric.py) evaluated ALONE by the engine (the
single-term kernel variants) against the golden per-term vectors (tests/golden/terms.npz), including
exact boundary values, nextafter neighbours, out-of-range points and the alpha = -1 branches; plus
B-spline projections against the golden design-matrix products (tests/golden/bases.npz)."""
import json
import os

import numpy as np
import pytest
from golden_util import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def _logw(pe_density, inj_density, theta_from):
    from gwinferno_amd.engine import NativePopulationLikelihood

    eng = NativePopulationLikelihood(pe_density, inj_density)
    th = eng.bound.theta_of(theta_from)
    lpe, linj = eng.log_weights(th)
    eng.close()
    return lpe, linj


def _check(got_log, ref_pdf, tol=1e-11):
    with np.errstate(all="ignore"):
        ref_log = np.log(ref_pdf)
    zero = ~(ref_pdf > 0) | ~np.isfinite(ref_pdf)  # zero, NaN or inf densities all mean "excluded"
    assert np.array_equal(np.isneginf(got_log), zero), (np.flatnonzero(np.isneginf(got_log) != zero)[:5])
    ok = ~zero
    assert np.max(np.abs(got_log[ok] - ref_log[ok])) < tol


def _pair(x):
    """(PE-shaped, injection-shaped) views of one test vector."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    return x.reshape(4, -1), x


@pytest.fixture(scope="module")
def terms():
    return np.load(os.path.join(GOLDEN_DIR, "terms.npz"))


def test_powerlaw_fixed_bounds(terms):
    from gwinferno_amd import models as M

    m1 = terms["m1"]
    pe, inj = _pair(m1)
    for tag, a in zip(("a", "b", "neg1", "zero"), terms["powerlaw_alphas"]):
        d_pe, d_inj = M.powerlaw_pdf(pe, a, 5.0, 100.0), M.powerlaw_pdf(inj, a, 5.0, 100.0)
        lpe, linj = _logw(d_pe, d_inj, d_pe)
        _check(linj, terms[f"powerlaw_pdf/{tag}"])
        _check(lpe.ravel(), terms[f"powerlaw_pdf/{tag}"])


def test_powerlaw_per_sample_lower_bound(terms):
    from gwinferno_amd import models as M

    m1, q = terms["m1"], terms["q"]
    for tag, b in zip(("a", "b", "neg1", "zero"), terms["powerlaw_alphas"]):
        qp, qi = _pair(q)
        with np.errstate(all="ignore"):
            lowp, lowi = _pair(5.0 / m1)
        d_pe, d_inj = M.powerlaw_pdf(qp, b, lowp, 1), M.powerlaw_pdf(qi, b, lowi, 1)
        lpe, linj = _logw(d_pe, d_inj, d_pe)
        _check(linj, terms[f"powerlaw_q/{tag}"])


def test_truncnorm_plpeak_and_ratio(terms):
    from gwinferno_amd import models as M

    m1, q = terms["m1"], terms["q"]
    mp, mi = _pair(m1)
    qp, qi = _pair(q)
    mu, sig, lo, hi = terms["truncnorm_params"]
    d_pe, d_inj = M.truncnorm_pdf(mp, mu, sig, lo, hi), M.truncnorm_pdf(mi, mu, sig, lo, hi)
    _check(_logw(d_pe, d_inj, d_pe)[1], terms["truncnorm_pdf"])
    mu, sig, lo, hi = terms["lognormal_params"]  # log=True: the log-normal branch (distributions.py:129-134)
    d_pe, d_inj = M.truncnorm_pdf(mp, mu, sig, lo, hi, log=True), M.truncnorm_pdf(mi, mu, sig, lo, hi, log=True)
    _check(_logw(d_pe, d_inj, d_pe)[1], terms["truncnorm_pdf_lognormal"])
    al, lo, hi, mpp, sigpp, lam = terms["plpeak_params"]
    d_pe, d_inj = M.plpeak_primary_pdf(mp, al, lo, hi, mpp, sigpp, lam), M.plpeak_primary_pdf(mi, al, lo, hi, mpp, sigpp, lam)
    _check(_logw(d_pe, d_inj, d_pe)[1], terms["plpeak_primary_pdf"])
    beta = float(terms["plpeak_ratio_beta"])
    d_pe = M.plpeak_primary_ratio_pdf(mp, qp, al, beta, lo, hi, mpp, sigpp, lam)
    d_inj = M.plpeak_primary_ratio_pdf(mi, qi, al, beta, lo, hi, mpp, sigpp, lam)
    _check(_logw(d_pe, d_inj, d_pe)[1], terms["plpeak_primary_ratio_pdf"])


def test_beta_and_tilt(terms):
    from gwinferno_amd import models as M

    ap, ai = _pair(terms["a"])
    a, b = terms["beta_params"]
    d_pe, d_inj = M.betadist(ap, a, b), M.betadist(ai, a, b)
    _check(_logw(d_pe, d_inj, d_pe)[1], terms["betadist"])
    a, b, scale = terms["beta_scaled_params"]  # Beta on [0, scale] (distributions.py:159-161)
    d_pe, d_inj = M.betadist(ap, a, b, scale=scale), M.betadist(ai, a, b, scale=scale)
    _check(_logw(d_pe, d_inj, d_pe)[1], terms["betadist_scaled"])
    cp, ci = _pair(terms["ct"])
    xi, sg = terms["tilt_params"]
    d_pe, d_inj = M.mixture_isoalign_spin_tilt(cp, xi, sg), M.mixture_isoalign_spin_tilt(ci, xi, sg)
    _check(_logw(d_pe, d_inj, d_pe)[1], terms["mixture_isoalign_spin_tilt"])


def test_redshift_model(terms):
    from gwinferno_amd import models as M

    zpe, zinj = terms["z_pe"], terms["z_inj"]
    zm = M.PowerlawRedshiftModel(zpe, zinj)
    assert np.array_equal([zm.zmin, zm.zmax], terms["z_model/zmin_zmax"])
    for i, lamb in enumerate(terms["z_lamb"]):
        d_pe, d_inj = zm(zpe, lamb), zm(zinj, lamb)
        lpe, linj = _logw(d_pe, d_inj, d_pe)
        _check(lpe, terms["z_model/pe"][i])
        _check(linj, terms["z_model/inj"][i])


def test_spline_projection_against_design_matrix():
    """Spline density at the golden sample points == the reference's dense projection (interpolation.py:306-317)
    for all four bases -- exp-splines (LogY / LogXLogY) and linear ones (BSpline / LogXBSpline, where a
    non-positive spline value counts as zero density) -- incl. domain ends and nextafter points."""
    from gwinferno_amd import models as M
    from gwinferno_amd.interpolation import BSpline, LogXBSpline, LogXLogYBSpline, LogYBSpline

    z = np.load(os.path.join(GOLDEN_DIR, "bases.npz"))
    meta = json.loads(str(z["meta"]))
    cls = {"LogYBSpline": LogYBSpline, "LogXLogYBSpline": LogXLogYBSpline, "BSpline": BSpline, "LogXBSpline": LogXBSpline}
    n_checked = 0
    for i, m in enumerate(meta):
        if m["cls"] not in cls:
            continue
        xs = z[f"{i}/xs"]
        xp, xi = _pair(xs)
        model = M.Base1DBSplineModel(m["n"], xp, xi, xrange=tuple(m["xrange"]), basis=cls[m["cls"]], normalize=True)
        cs = z[f"{i}/coefs"]
        if not cls[m["cls"]].log_y and z[f"{i}/norm"] < 0:
            # random coefficients of either sign can make the integral of a LINEAR spline negative; the reference's
            # f / Z is then positive where f < 0.  -c describes the same normalised density with f, Z > 0, the
            # only case the engine defines (a non-positive spline value is zero density)
            cs = -cs
        d_pe, d_inj = model(cs, pe_samples=True), model(cs, pe_samples=False)
        lpe, linj = _logw(d_pe, d_inj, d_pe)
        _check(linj, z[f"{i}/project"], tol=1e-10)
        n_checked += 1
    assert n_checked == 9


def test_numpyro_distribution_log_probs(terms):
    """Powerlaw (bounds as hyper-parameters: samples ON the bounds and their nextafter neighbours, alpha = -1),
    PowerlawRedshift and BSplineDistribution on the four bases, each evaluated ALONE by the engine, against the
    reference's own log_prob arrays (tests/golden/terms.npz dist/*; numpyro_distributions.py:127-136, 186-195, 296-301)."""
    from gwinferno_amd import interpolation as I
    from gwinferno_amd import numpyro_distributions as D

    def check(got, ref, tol=1e-11):
        dead = ref < -1e300  # nan_to_num(-inf)
        assert np.array_equal(np.isneginf(got), dead)
        assert np.max(np.abs(got[~dead] - ref[~dead])) < tol

    x = terms["m1"]
    pe, inj = _pair(x)
    for tag, a in zip(("a", "b", "neg1", "zero"), terms["powerlaw_alphas"]):
        d_pe, d_inj = D.Powerlaw(a, 5.0, 100.0).log_prob(pe), D.Powerlaw(a, 5.0, 100.0).log_prob(inj)
        lpe, linj = _logw(d_pe, d_inj, d_pe)
        check(linj, terms[f"dist/powerlaw/{tag}"])
        check(lpe.ravel(), terms[f"dist/powerlaw/{tag}"])
    zg, dv = terms["dist/z_grid"], terms["dist/z_dVcdz"]
    zpe, zinj = _pair(terms["z_inj"])
    for i, lamb in enumerate(terms["z_lamb"]):
        mk = lambda: D.PowerlawRedshift(lamb, float(terms["dist/powerlaw_redshift/maximum"]), zg, dv)  # noqa: E731
        d_pe, d_inj = mk().log_prob(zpe), mk().log_prob(zinj)
        lpe, linj = _logw(d_pe, d_inj, d_pe)
        check(linj, terms["dist/powerlaw_redshift/inj"][i])
        check(lpe.ravel(), terms["dist/powerlaw_redshift/inj"][i])
    v, cs = terms["dist/bspline/value"], terms["dist/bspline/cs"]
    vpe, vinj = _pair(v)
    gr, grx = np.linspace(0, 1, 1000), np.linspace(0.001, 1, 1000)
    for tag, basis, g in (("bspline", I.BSpline(20, normalize=True), gr), ("logy", I.LogYBSpline(20, normalize=True), gr), ("logx", I.LogXBSpline(20, normalize=True), grx),
                          ("logxy", I.LogXLogYBSpline(20, xrange=(0.001, 1), normalize=True), grx)):
        dm = basis.bases(g)
        mk = lambda: D.BSplineDistribution(g[0], g[-1], cs, g, dm)  # noqa: E731
        d_pe, d_inj = mk().log_prob(vpe), mk().log_prob(vinj)
        lpe, linj = _logw(d_pe, d_inj, d_pe)
        check(linj, terms[f"dist/bspline/{tag}"])
        check(lpe.ravel(), terms[f"dist/bspline/{tag}"])


def test_bspline_distribution_grid_reaching_outside_a_log_y_basis():
    """A grid that starts below the domain of a log-Y basis (the default LogXLogYBSpline(20) on a grid from 0.001,
    tests/numpyro_distributions_test.py:126-129): those grid points carry lpdf = -inf, so samples interpolating
    into them have zero weight, the others follow the NumPy statement of the same table (tests/bound_eval.py)."""
    from bound_eval import log_weights

    from gwinferno_amd import interpolation as I
    from gwinferno_amd import numpyro_distributions as D
    from gwinferno_amd.engine import NativePopulationLikelihood

    rng = np.random.default_rng(5)
    g = np.linspace(0.001, 1, 1000)
    dm = I.LogXLogYBSpline(20, normalize=True).bases(g)  # domain (0.1, 1)
    cs = rng.normal(size=20)
    v = rng.uniform(0.0, 1.05, 4096)
    vpe, vinj = _pair(v)
    mk = lambda: D.BSplineDistribution(0.001, 1.0, cs, g, dm)  # noqa: E731
    d_pe, d_inj = mk().log_prob(vpe), mk().log_prob(vinj)
    eng = NativePopulationLikelihood(d_pe, d_inj)
    th = eng.bound.theta_of(d_pe)
    lpe, linj = eng.log_weights(th)
    rpe, rinj, _ = log_weights(eng.bound, th)
    eng.close()
    assert np.isneginf(linj).sum() > 300 and np.isfinite(linj).sum() > 3000
    for got, ref in ((lpe, rpe), (linj, rinj)):
        assert np.array_equal(np.isneginf(got), np.isneginf(ref))
        ok = np.isfinite(ref)
        assert np.max(np.abs(got[ok] - ref[ok])) < 1e-11
    below = vinj < 0.1 - (g[1] - g[0])
    assert np.all(np.isneginf(linj[below]))
