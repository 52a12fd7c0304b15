"""GPU (-m gpu): randomised parity sweep -- every composition, many hyper-points drawn WIDER than the
benchmark priors (steep slopes, narrow / wide peaks, mixing fractions near 0 and 1, large spline
coefficients), HIP engine vs the C oracle (itself pinned to the reference's golden vectors, tests/test_c_oracle.py)
on one seeded mid-size catalog.  Values to 1e-9, analytic gradients to 1e-8 of their scale."""
import os

import numpy as np
import pytest
from golden_util import rel_err

pytestmark = pytest.mark.gpu

N_POINTS = int(os.environ.get("GWI_FUZZ_POINTS", "48"))  # the long sweep: GWI_FUZZ_POINTS=400


def _wide(name, p, rng):
    """Push a benchmark draw outwards (in place)."""
    for k, v in p.items():
        if np.ndim(v) > 0:
            p[k] = np.asarray(v) * rng.uniform(0.5, 4.0)  # spline coefficients up to ~ +-10
            if k == "z_coefs" and name != "bspline_redshift":
                p[k][0] = 0.0
        elif k in ("alpha", "beta", "lamb"):
            p[k] = float(v) + rng.normal(0.0, 3.0)
        elif k in ("sigpp",):
            p[k] = float(rng.uniform(0.3, 25.0))
        elif k in ("lam", "xi", "xi1", "xi2"):
            p[k] = float(rng.choice([rng.uniform(0, 1), rng.uniform(0, 1e-3), 1 - rng.uniform(0, 1e-3)]))
        elif k.startswith("sig_t"):
            p[k] = float(rng.uniform(0.05, 6.0))
        elif k == "delta":
            p[k] = float(rng.uniform(0.1, 30.0))
        elif k == "mmin":  # truncation bounds as hyper-parameters: cutting into the data from both ends
            p[k] = float(rng.uniform(1.0, 12.0))
        elif k == "mmax":
            p[k] = float(rng.uniform(60.0, 130.0))
    if name == "bspline_redshift":  # exponent coefficients are c / (c . I): keep the denominator away from 0
        p["z_coefs"] = np.abs(p["z_coefs"]) + 0.05
    for k in ("e_coefs", "p_coefs"):  # linear (density) splines need positive coefficients
        if k in p:
            p[k] = np.abs(p[k]) + 0.01
    return p


@pytest.mark.parametrize("name", ["pl_test", "plpeak", "plpeak_full", "plpeak_default_tilt", "bspline_test", "bspline_iid", "bspline_full", "bspline_defaults", "bspline_chieff",
                                  "bspline_component_masses", "bspline_redshift", "bspline_redshift_raw", "plpeak_smooth", "chm_powerlaw", "chm_bspline", "bspline_misc", "bspline_independent_masses", "plpeak_iid_spins"])
def test_randomised_parity_against_c_oracle(name):
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle.c_oracle import COracle

    pe, inj, total = make_catalog(14, 600, 6000, seed=77)
    comp = COMPOSITIONS[name](pe, inj)
    eng = comp.engine()
    orc = COracle(eng.bound)
    rng = np.random.default_rng(sum(map(ord, name)))
    worst = 0.0
    n_finite = 0
    pending = []
    for i in range(N_POINTS):
        p = draw_params(name, rng)
        if name == "chm_powerlaw":  # bounds that cut samples off but leave every event some: the catalog's events span m1 ~ 8 .. 80
            p["mmin"], p["mmax"] = float(rng.uniform(1.0, 6.5)), float(rng.uniform(88.0, 130.0))
        if i >= N_POINTS // 3:
            p = _wide(name, p, rng)
        if name == "chm_powerlaw" and i % 2:
            p["mmin"], p["mmax"] = float(rng.uniform(1.0, 6.5)), float(rng.uniform(88.0, 130.0))
        th = comp.theta(p)
        got = eng.evaluate(th, total, min_neff_cut=False)
        ref = orc.evaluate(th, total, min_neff_cut=False)
        rs = ref["summary"]
        ok = np.isfinite(ref["logBFs"])
        assert np.array_equal(np.isfinite(got.log_bfs), ok), (name, i)
        assert rel_err(got.log_bfs[ok], ref["logBFs"][ok]) < 1e-9, (name, i, p)
        # log n_eff = 2 log S1 - log S2 cancels when ONE sample carries an event (n_eff -> 1: a 0.4-wide peak holding 99.99 % of
        # the mixture gave log n_eff = 2.8e-10, on which this engine, the C oracle and the NumPy oracle all differ at 1e-6
        # relative): the error is measured against max(1, |log n_eff|)
        d_neff = np.abs(got.log_neffs[ok] - ref["log_nEffs"][ok]) / np.maximum(1.0, np.abs(ref["log_nEffs"][ok]))
        assert d_neff.size == 0 or float(np.max(d_neff)) < 1e-8, (name, i)
        if np.isfinite(rs.log_det_eff):
            assert abs(got.summary.log_det_eff - rs.log_det_eff) < 1e-9 * max(1.0, abs(rs.log_det_eff)), (name, i)
        assert got.log_likelihood == rs.log_likelihood or rel_err(got.log_likelihood, rs.log_likelihood) < 1e-9, (name, i, got.log_likelihood, rs.log_likelihood)
        if abs(rs.log_likelihood) < 1e300:
            n_finite += 1
            scale = max(1.0, float(np.max(np.abs(ref["grad"]))))
            err = float(np.max(np.abs(got.grad - ref["grad"]))) / scale
            worst = max(worst, err)
            assert err < 1e-8, (name, i, err)
        # every 8th point also with the selection uncertainty marginalised (analysis.py:270-271; the gradient then takes the
        # squared-weight pass of the scan, KArgs::square) or with one of the cuts on (min_neff_cut / max_variance_cut), by turns
        if i % 8 == 3:
            flags = dict(marginalize_selection=True, min_neff_cut=False) if i % 16 == 3 else dict(min_neff_cut=True) if i % 32 == 11 else dict(min_neff_cut=False, max_variance_cut=True)
            gm = eng.evaluate(th, total, **flags)
            rm = orc.evaluate(th, total, **flags)
            rms = rm["summary"]
            assert gm.log_likelihood == rms.log_likelihood or rel_err(gm.log_likelihood, rms.log_likelihood) < 1e-9, (name, i, flags, gm.log_likelihood, rms.log_likelihood)
            if abs(rms.log_likelihood) < 1e300:
                assert np.all(np.isfinite(gm.grad)), (name, i, flags)
                # value AND gradient: the C oracle carries the gradient of the marginalised selection term too (pinned to the
                # reference's finite differences by tests/test_c_oracle.py)
                sc = max(1.0, float(np.max(np.abs(rm["grad"]))))
                assert float(np.max(np.abs(gm.grad - rm["grad"]))) / sc < 1e-8, (name, i, flags)
        # the same points eight at a time through the batched launch (its own kernel instantiation and tail): equal to the
        # single evaluations up to summation-order rounding
        pending.append((th, got))
        if len(pending) == 8:
            batch = eng.evaluate_batch(np.stack([t for t, _ in pending]), total, min_neff_cut=False)
            for k, (_, single) in enumerate(pending):
                b = batch[k]
                assert b.log_likelihood == single.log_likelihood or rel_err(b.log_likelihood, single.log_likelihood) < 1e-11, (name, i, k)
                if abs(single.log_likelihood) < 1e300:
                    sc = max(1.0, float(np.max(np.abs(single.grad))))
                    assert float(np.max(np.abs(b.grad - single.grad))) / sc < 1e-10, (name, i, k)
            pending.clear()
    assert n_finite >= N_POINTS // 2, (name, n_finite)  # the sweep must mostly land on live likelihoods
    eng.close()


@pytest.mark.parametrize("name", ["plpeak_smooth", "plpeak"])
def test_narrow_peak_gradient_stays_finite(name):
    """Found by the 2000-point sweep (GWI_FUZZ_POINTS=2000): a 0.33-wide Gaussian peak holding 0.09 % of the mixture.  The
    mixture term carries the sample's closing factor e^{l - m} inside its density (scan_kernel, Absorbs<K>), so samples ~700
    e-folds under the tile's best one present SUBNORMAL densities, whose reciprocal (fast_rcp: hardware seed + Newton) is NaN;
    their gradient states must count as zero (kRcpFloor) instead of poisoning the sums."""
    from gwinferno_amd.compositions import COMPOSITIONS
    from gwinferno_amd.synthetic import make_catalog
    from oracle.c_oracle import COracle

    pe, inj, total = make_catalog(14, 600, 6000, seed=77)
    comp = COMPOSITIONS[name](pe, inj)
    eng = comp.engine()
    orc = COracle(eng.bound)
    p = {"alpha": -2.8556157475292885, "beta": -1.0778941566544196, "mpp": 39.23571240555825, "sigpp": 0.3296513261163328, "lam": 0.0008906616210867098, "lamb": 5.088845939224041}
    if name == "plpeak_smooth":
        p["delta"] = 9.250712517195849
    for sig in (0.3296513261163328, 0.12, 0.05):
        p["sigpp"] = sig
        th = comp.theta(p)
        got = eng.evaluate(th, total, min_neff_cut=False)
        ref = orc.evaluate(th, total, min_neff_cut=False)
        assert np.all(np.isfinite(got.grad)), (name, sig, got.grad)
        scale = max(1.0, float(np.max(np.abs(ref["grad"]))))
        assert float(np.max(np.abs(got.grad - ref["grad"]))) / scale < 1e-8, (name, sig)
        assert rel_err(got.log_likelihood, ref["summary"].log_likelihood) < 1e-9
    eng.close()


N_SHAPES = int(os.environ.get("GWI_FUZZ_SHAPES", "24"))  # the long sweep: GWI_FUZZ_SHAPES=400


@pytest.mark.parametrize("name", ["plpeak", "bspline_iid", "bspline_misc"])
def test_randomised_catalog_shapes_against_c_oracle(name, monkeypatch):
    """Random catalog shapes (events x posterior samples x injections, from a single sample per event to tens of thousands)
    and, for a third of them, a random tile size: the tile / group bookkeeping of the scan and its tail against the C oracle
    -- single evaluations and a 5-point batched launch."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle.c_oracle import COracle

    rng = np.random.default_rng(sum(map(ord, name)) + 17)
    for s in range(N_SHAPES):
        n_ev = int(rng.choice([1, 2, 3, 7, 8, 9, 33, 64, 65, int(rng.integers(1, 300))]))
        n_pe = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 513, int(rng.integers(1, 3000))]))
        n_inj = int(rng.choice([1, 63, 64, 65, 257, 4097, int(rng.integers(1, 20000))]))
        env = {}
        if s % 3 == 2:
            env["GWI_SAMPLES_PER_BLOCK"] = str(int(rng.choice([256, 512, 768, 1024, 4096])))
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pe, inj, total = make_catalog(n_ev, n_pe, n_inj, seed=1000 + s)
        comp = COMPOSITIONS[name](pe, inj)
        eng = comp.engine()
        orc = COracle(eng.bound)
        thetas = np.stack([comp.theta(draw_params(name, rng)) for _ in range(5)])
        batch = eng.evaluate_batch(thetas, total, min_neff_cut=False)
        for k in range(2):
            got = eng.evaluate(thetas[k], total, min_neff_cut=False)
            ref = orc.evaluate(thetas[k], total, min_neff_cut=False)
            tag = (name, s, n_ev, n_pe, n_inj, env)
            ok = np.isfinite(ref["logBFs"])
            assert np.array_equal(np.isfinite(got.log_bfs), ok), tag
            assert rel_err(got.log_bfs[ok], ref["logBFs"][ok]) < 1e-9, tag
            rs = ref["summary"]
            assert got.log_likelihood == rs.log_likelihood or rel_err(got.log_likelihood, rs.log_likelihood) < 1e-9, tag
            if abs(rs.log_likelihood) < 1e300:
                scale = max(1.0, float(np.max(np.abs(ref["grad"]))))
                assert float(np.max(np.abs(got.grad - ref["grad"]))) / scale < 1e-8, tag
                assert float(np.max(np.abs(batch[k].grad - ref["grad"]))) / scale < 1e-8, tag
                assert rel_err(batch[k].log_likelihood, rs.log_likelihood) < 1e-9, tag
        eng.close()
        for k in env:
            monkeypatch.delenv(k)
