"""CPU: gwinferno_amd.pipeline_utils (the factories and prior helpers of gwinferno/pipeline/utils.py:104-216) against
tests/golden/pipeline.npz, produced by running the reference's own functions (tests/golden/make_golden.py pipeline)."""
import json
import os

import numpy as np
import pytest
from bound_eval import log_weights
from golden_util import GOLDEN_DIR

from gwinferno_amd import likelihood as L
from gwinferno_amd import pipeline_utils as U
from gwinferno_amd.engine import bind
from gwinferno_amd.synthetic import make_catalog


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(GOLDEN_DIR, "pipeline.npz"))


def _inject(fx):
    for k in fx.files:
        if k.startswith("sample/"):
            L.SAMPLE_VALUES[k[7:]] = fx[k]


def test_prior_helpers_reproduce_the_reference_factors(fx):
    ns = json.loads(str(fx["meta"]))["nsplines"]
    _inject(fx)
    U.PRIOR_FACTORS.clear()
    mass_cs, q_cs = U.bspline_mass_prior(m_nsplines=ns["m1"], q_nsplines=ns["q"], m_tau=1, q_tau=1)
    a1, t1, a2, t2 = U.bspline_spin_prior(a_nsplines=ns["a"], ct_nsplines=ns["ct"], a_tau=25, ct_tau=25, IID=False)
    z_cs = U.bspline_redshift_prior(z_nsplines=ns["z"], z_tau=1)
    U.bspline_spin_prior(a_nsplines=ns["a"], ct_nsplines=ns["ct"], a_tau=3.0, ct_tau=0.5, IID=True, name="tag", a_deg=1, ct_deg=3)
    only_q = U.bspline_mass_prior(q_nsplines=ns["q"], q_tau=7.0, q_deg=2)
    want = {k[7:]: float(fx[k]) for k in fx.files if k.startswith("factor/")}
    assert set(U.PRIOR_FACTORS) == set(want)
    for k, v in want.items():
        assert abs(U.PRIOR_FACTORS[k] - v) <= 1e-13 * abs(v), k
    assert np.array_equal(z_cs, fx["returned/z_cs"]) and z_cs[0] == 0 and np.array_equal(only_q, fx["returned/only_q"])
    assert np.array_equal(mass_cs, fx["sample/mass_cs"]) and np.array_equal(a2, fx["sample/a2_cs"]) and np.array_equal(t1, fx["sample/tilt1_cs"])
    with pytest.raises(AssertionError):
        U.bspline_mass_prior()


def _example_product(fx):
    meta = json.loads(str(fx["meta"]))
    ns = meta["nsplines"]
    n_ev, n_pe, n_inj, seed = meta["catalog"]
    pe, inj, _ = make_catalog(n_ev, n_pe, n_inj, seed=seed)
    mass_models = U.setup_bspline_mass_models(pe, inj, ns["m1"], ns["q"], meta["mmin"], meta["mmax"])
    mag_model, tilt_model = U.setup_bspline_spin_models(pe, inj, ns["a"], ns["ct"], IID=False, a2_nsplines=ns["a"], ct2_nsplines=ns["ct"])
    z_model = U.setup_powerlaw_spline_redshift_model(pe, inj, ns["z"])
    cs = {k[7:]: fx[k] for k in fx.files if k.startswith("sample/")}
    z_cs, lamb = np.concatenate([np.zeros(1), cs["z_cs"]]), float(fx["lamb"])

    def weights(d, flag):  # examples/simple_bspline_example.py:60-71
        return (mass_models(cs["mass_cs"], cs["q_cs"], pe_samples=flag) * mag_model(cs["a1_cs"], cs["a2_cs"], pe_samples=flag)
                * tilt_model(cs["tilt1_cs"], cs["tilt2_cs"], pe_samples=flag) * z_model(d["redshift"], lamb, z_cs) / d["prior"])

    return weights(pe, True), weights(inj, False), z_model.normalization(lamb, z_cs)


def test_factories_build_the_reference_models(fx):
    wp, wi, hv = _example_product(fx)
    bm = bind(wp, wi, hv)
    lpe, linj, norms = log_weights(bm, bm.theta_of(wp))
    for got, ref in ((lpe, fx["factory/pe"]), (linj, fx["factory/inj"])):
        with np.errstate(all="ignore"):
            rl = np.log(ref)
        dead = ~(ref > 0)
        assert np.array_equal(np.isneginf(got), dead)
        assert np.max(np.abs(got[~dead] - rl[~dead])) < 1e-10
    assert abs(norms[bm.vt_norm] / float(fx["factory/hypervolume"]) - 1) < 1e-12


def test_example_prior_for_the_native_sampler(fx):
    """bspline_example_prior == the sum of the reference's Normal log-densities (up to their constants) and factor
    sites; the pinned redshift coefficient goes through a FIXED bijector slot."""
    ns = json.loads(str(fx["meta"]))["nsplines"]
    order = [("m1", ns["m1"]), ("q", ns["q"]), ("a1", ns["a"]), ("a2", ns["a"]), ("tilt1", ns["ct"]), ("tilt2", ns["ct"]), ("redshift", ns["z"]), ("lamb", 1)]
    slices, off = {}, 0
    for k, n in order:
        slices[k] = slice(off, off + n)
        off += n
    prior, bij = U.bspline_example_prior(slices)
    cs = {k[7:]: fx[k] for k in fx.files if k.startswith("sample/")}
    theta = np.concatenate([cs["mass_cs"], cs["q_cs"], cs["a1_cs"], cs["a2_cs"], cs["tilt1_cs"], cs["tilt2_cs"], np.zeros(1), cs["z_cs"], [1.3]])
    lp, grad = prior(theta)
    normal = -0.5 * (np.sum(cs["mass_cs"] ** 2) / 15**2 + np.sum(cs["q_cs"] ** 2) / 5**2 + sum(np.sum(cs[k] ** 2) for k in ("a1_cs", "a2_cs", "tilt1_cs", "tilt2_cs")) / 5**2
                     + np.sum(cs["z_cs"] ** 2) / 1**2 + 1.3**2 / 3**2)
    _inject(fx)
    U.PRIOR_FACTORS.clear()
    U.bspline_mass_prior(m_nsplines=ns["m1"], q_nsplines=ns["q"], m_tau=1, q_tau=1)
    U.bspline_spin_prior(a_nsplines=ns["a"], ct_nsplines=ns["ct"], a_tau=25, ct_tau=25, IID=False)
    U.bspline_redshift_prior(z_nsplines=ns["z"], z_tau=1)
    assert abs(lp - (normal + sum(U.PRIOR_FACTORS.values()))) < 1e-10 * abs(lp)
    zi = slices["redshift"].start
    u = bij.inverse(theta)
    th, dth, dlogj, logj = bij.forward(u + 0.7)  # moving the dummy coordinate does not move the pinned parameter
    assert bij.kind[zi] == 3 and th[zi] == 0.0 and dth[zi] == 0.0 and np.allclose(np.delete(th, zi), np.delete(theta, zi) + 0.7)
    assert abs(logj + 0.5 * 0.7**2) < 1e-15 and dlogj[zi] == -0.7
