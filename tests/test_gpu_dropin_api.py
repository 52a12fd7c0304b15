"""GPU (-m gpu): the reference's own integration tests (tests/inference_test.py:162-347) restated
against the drop-in API: the model functions below read like the reference's, with
gwinferno_amd.models / gwinferno_amd.likelihood in place of gwinferno.models / pipeline.analysis.
The reference only asserts finiteness of value and gradient at its test points; here the value and
every site are additionally compared with the golden vectors produced by the reference itself."""
import numpy as np
import pytest
from golden_util import GoldenCase, rel_err

pytestmark = pytest.mark.gpu


def _parametric_model(case):
    """tests/inference_test.py:162-197."""
    from gwinferno_amd.lazy import where_finite
    from gwinferno_amd.likelihood import hierarchical_likelihood
    from gwinferno_amd.models import PowerlawRedshiftModel, powerlaw_primary_ratio_pdf

    pedict, injdict = case.pe, case.inj
    z_model = PowerlawRedshiftModel(z_pe=pedict["redshift"], z_inj=injdict["redshift"])
    mmin, mmax = case.meta["mmin"], case.meta["mmax"]

    def model(alpha, beta, lamb, **flags):
        def get_weights(m1, q, z, prior):
            p_m1q = powerlaw_primary_ratio_pdf(m1, q, alpha=alpha, beta=beta, mmin=mmin, mmax=mmax)
            p_z = z_model(z, lamb)
            wts = p_m1q * p_z / prior
            return where_finite(wts)

        peweights = get_weights(pedict["mass_1"], pedict["mass_ratio"], pedict["redshift"], pedict["prior"])
        injweights = get_weights(injdict["mass_1"], injdict["mass_ratio"], injdict["redshift"], injdict["prior"])
        return hierarchical_likelihood(peweights, injweights, total_inj=case.total_inj, Nobs=case.nobs, Tobs=case.tobs,
                                       surveyed_hypervolume=z_model.normalization(lamb=lamb), marginalize_selection=False, **flags)

    return model


def _bspline_model(case):
    """tests/inference_test.py:124-143, 244-285."""
    from gwinferno_amd.lazy import where_finite
    from gwinferno_amd.likelihood import hierarchical_likelihood
    from gwinferno_amd.models import BSplinePrimaryBSplineRatio, PowerlawSplineRedshiftModel

    pedict, injdict = case.pe, case.inj
    mmin, mmax = case.meta["mmin"], case.meta["mmax"]
    mass_model = BSplinePrimaryBSplineRatio(10, 5, pedict["mass_1"], injdict["mass_1"], pedict["mass_ratio"], injdict["mass_ratio"], m1min=mmin, m2min=mmin, mmax=mmax)
    z_model = PowerlawSplineRedshiftModel(5, pedict["redshift"], injdict["redshift"])

    def model(m1_coefs, q_coefs, z_coefs, lamb, **flags):
        def get_weights(z, prior, pe_samples=False):
            p_m1q = mass_model(m1_coefs, q_coefs, pe_samples=pe_samples)
            p_z = z_model(z, lamb, z_coefs)
            return where_finite(p_m1q * p_z / prior)

        peweights = get_weights(pedict["redshift"], pedict["prior"], pe_samples=True)
        injweights = get_weights(injdict["redshift"], injdict["prior"], pe_samples=False)
        return hierarchical_likelihood(peweights, injweights, total_inj=case.total_inj, Nobs=case.nobs, Tobs=case.tobs,
                                       surveyed_hypervolume=z_model.normalization(lamb=lamb, cs=z_coefs), marginalize_selection=False, **flags)

    return model


@pytest.mark.parametrize("name,builder", [("pl_test", _parametric_model), ("gwtc3_pl_test", _parametric_model), ("bspline_test", _bspline_model), ("gwtc3_bspline_test", _bspline_model)])
def test_reference_style_model_functions(name, builder):
    from gwinferno_amd import likelihood as L

    case = GoldenCase(name)
    model = builder(case)
    L.SAMPLE_VALUES["unscaled_rate"] = case.meta["unscaled_rate"]
    for fs in ("lin", "lin_neff"):
        flags = {k: v for k, v in case.flagsets[fs].items() if k != "log"}
        for i in range(case.n_points):
            rate = model(**case.point(i), **flags)
            sites = L.last_sites()
            # tests/inference_test.py:328-329, 346-347
            assert np.isfinite(sites["log_likelihood"]) and np.all(np.isfinite(sites["grad_log_likelihood"]))
            for site, ref in case.sites[fs].items():
                if site == "rate_return":
                    got = rate
                else:
                    got = sites[site]
                if site.startswith("variance"):
                    assert np.allclose(got, ref[i], rtol=1e-8, atol=1e-12), (name, fs, i, site)
                else:
                    assert rel_err(got, ref[i]) < 1e-9, (name, fs, i, site, got, ref[i])
    L.clear_engine_cache()


def _weights(name, case, point):
    """PE and injection products of the two reference test models (inference_test.py:168-175, 255-262)."""
    from gwinferno_amd.lazy import where_finite
    from gwinferno_amd.models import BSplinePrimaryBSplineRatio, PowerlawRedshiftModel, PowerlawSplineRedshiftModel, powerlaw_primary_ratio_pdf

    pe, inj = case.pe, case.inj
    mmin, mmax = case.meta["mmin"], case.meta["mmax"]
    if "bspline" in name:
        mass_model = BSplinePrimaryBSplineRatio(10, 5, pe["mass_1"], inj["mass_1"], pe["mass_ratio"], inj["mass_ratio"], m1min=mmin, m2min=mmin, mmax=mmax)
        z_model = PowerlawSplineRedshiftModel(5, pe["redshift"], inj["redshift"])
        w = lambda d, flag: where_finite(mass_model(point["m1_coefs"], point["q_coefs"], pe_samples=flag) * z_model(d["redshift"], point["lamb"], point["z_coefs"]) / d["prior"])  # noqa: E731
        return w(pe, True), w(inj, False)
    z_model = PowerlawRedshiftModel(z_pe=pe["redshift"], z_inj=inj["redshift"])
    w = lambda d: where_finite(powerlaw_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], alpha=point["alpha"], beta=point["beta"], mmin=mmin, mmax=mmax)  # noqa: E731
                               * z_model(d["redshift"], point["lamb"]) / d["prior"])
    return w(pe), w(inj)


@pytest.mark.parametrize("name", ["pl_test", "bspline_test"])
def test_one_sided_reference_functions(name):
    """per_event_log_bayes_factors / detection_efficiency (analysis.py:50-136) called on their own with
    a lazy product, as a user's diagnostics code would: same arrays as the golden sites."""
    from gwinferno_amd import likelihood as L

    case = GoldenCase(name)
    ref = case.sites["lin"]
    for i in range(min(case.n_points, 4)):
        pew, injw = _weights(name, case, case.point(i))
        log_bfs, log_neffs, variances = L.per_event_log_bayes_factors(pew)
        assert rel_err(log_bfs, ref["logBFs"][i]) < 1e-9
        assert rel_err(log_neffs, ref["log_nEffs"][i]) < 1e-9
        assert np.allclose(variances, ref["variance_log_BFs"][i], rtol=1e-8, atol=1e-12)
        logmu, logneff, var = L.detection_efficiency(injw, case.total_inj)
        assert rel_err(np.exp(logmu), ref["detection_efficiency"][i]) < 1e-9
        assert rel_err(logneff, ref["log_nEff_inj"][i]) < 1e-9
        assert np.allclose(var, ref["variance_log_detection_efficiency"][i], rtol=1e-8, atol=1e-12)
    with pytest.raises(ValueError):
        L.per_event_log_bayes_factors(injw)
    with pytest.raises(ValueError):
        L.detection_efficiency(pew, case.total_inj)
    L.clear_engine_cache()


def test_engine_is_cached_across_calls():
    from gwinferno_amd import likelihood as L

    case = GoldenCase("pl_test")
    model = _parametric_model(case)
    model(**case.point(0), min_neff_cut=False)
    n = len(L._ENGINES)
    model(**case.point(1), min_neff_cut=False)
    assert len(L._ENGINES) == n == 1
    L.clear_engine_cache()


def test_builtin_hmc_runs_end_to_end():
    """Prior + engine likelihood + HMC (gwinferno_amd.sampling): finite, moving chain with a sane
    acceptance rate -- the counterpart of the reference's skipped 5+5-step NUTS smoke tests
    (tests/inference_test.py:367-411)."""
    from gwinferno_amd.compositions import COMPOSITIONS
    from gwinferno_amd.sampling import GaussianSmoothingPrior, hmc, make_target
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(20, 400, 4000, seed=3)
    comp = COMPOSITIONS["pl_test"](pe, inj)
    eng = comp.engine()
    theta0 = comp.theta({"alpha": -2.0, "beta": 1.0, "lamb": 2.0})
    prior = GaussianSmoothingPrior(eng.n_theta).normal(slice(0, eng.n_theta), 5.0)
    out = hmc(make_target(eng, total, prior, min_neff_cut=False), theta0, n_warmup=60, n_samples=40, n_leapfrog=6, seed=2)
    assert out["samples"].shape == (40, eng.n_theta)
    assert np.all(np.isfinite(out["samples"])) and np.all(np.isfinite(out["log_prob"]))
    assert 0.3 < out["accept_rate"] <= 1.0
    assert np.std(out["samples"], axis=0).min() > 0  # the chain moves
    eng.close()


def test_builtin_nuts_runs_end_to_end():
    """The reference's sampler flow (NUTS over the model, examples/utils.py:63-85) with the built-in driver:
    engine likelihood + priors, a short run that moves, accepts and stays finite; every leapfrog step is one
    engine evaluation (counted).  Same catalog and model as the HMC test above."""
    from gwinferno_amd.compositions import COMPOSITIONS
    from gwinferno_amd.sampling import GaussianSmoothingPrior, make_target, nuts
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(20, 400, 4000, seed=3)
    comp = COMPOSITIONS["pl_test"](pe, inj)
    eng = comp.engine()
    theta0 = comp.theta({"alpha": -2.0, "beta": 1.0, "lamb": 2.0})
    prior = GaussianSmoothingPrior(eng.n_theta).normal(slice(0, eng.n_theta), 5.0)
    out = nuts(make_target(eng, total, prior, min_neff_cut=False), theta0, n_warmup=80, n_samples=60, seed=4, max_tree_depth=6)
    assert out["samples"].shape == (60, eng.n_theta)
    assert np.all(np.isfinite(out["samples"])) and np.all(np.isfinite(out["log_prob"]))
    assert 0.4 < out["accept_rate"] <= 1.0 and out["n_divergent"] < 10
    assert np.std(out["samples"], axis=0).min() > 0
    assert out["n_evals"] >= 140 and out["tree_depth"].max() <= 6
    eng.close()


def test_interleaved_nuts_chains_on_one_gpu():
    """Three NUTS chains, one engine each, evaluations in flight together (gwi_eval_begin / gwi_eval_end): every
    chain equals the same chain run alone through the blocking entry."""
    from gwinferno_amd.compositions import COMPOSITIONS
    from gwinferno_amd.sampling import GaussianSmoothingPrior, make_async_target, make_target, nuts, nuts_chains
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(20, 400, 4000, seed=3)
    comps = [COMPOSITIONS["pl_test"](pe, inj) for _ in range(3)]
    engs = [c.engine() for c in comps]
    theta0 = comps[0].theta({"alpha": -2.0, "beta": 1.0, "lamb": 2.0})
    prior = GaussianSmoothingPrior(engs[0].n_theta).normal(slice(0, engs[0].n_theta), 5.0)
    starts = [theta0, theta0 + 0.1, theta0 - 0.1]
    res = nuts_chains([make_async_target(e, total, prior, min_neff_cut=False) for e in engs], starts, n_warmup=30, n_samples=20, seed=7, max_tree_depth=5)
    for c, r in enumerate(res):
        alone = nuts(make_target(engs[0], total, prior, min_neff_cut=False), starts[c], n_warmup=30, n_samples=20, seed=7 + 1000 * c, max_tree_depth=5)
        assert np.allclose(r["samples"], alone["samples"], rtol=1e-12, atol=1e-12) and r["n_evals"] == alone["n_evals"]
        assert np.all(np.isfinite(r["log_prob"]))
    for e in engs:
        e.close()


def _native_nuts_setup(n_engines):
    from gwinferno_amd.compositions import COMPOSITIONS
    from gwinferno_amd.sampling import Bijector, GaussianSmoothingPrior
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(20, 400, 4000, seed=3)
    comps = [COMPOSITIONS["pl_test"](pe, inj) for _ in range(n_engines)]
    engs = [c.engine() for c in comps]
    n = engs[0].n_theta
    theta0 = comps[0].theta({"alpha": -2.0, "beta": 1.0, "lamb": 2.0})
    # every feature of the native target at once: a Normal prior, an interval and a positive bijector and a
    # (physically meaningless here, arithmetically complete) first-difference penalty over all parameters
    prior = GaussianSmoothingPrior(n).normal(slice(0, n), 5.0).smoothing(slice(0, n), 0.05, 1)
    names = list(comps[0].names) if hasattr(comps[0], "names") else None
    bij = Bijector(n)
    order = np.argsort(theta0)  # most negative entry gets the interval, most positive the positive map
    bij.interval(int(order[0]), theta0[order[0]] - 4.0, theta0[order[0]] + 3.0).positive(int(order[-1]))
    return engs, total, prior, bij, theta0, names


def test_native_nuts_on_the_engine_matches_the_callback_path():
    """gwi_nuts_engine (target assembled in C++: likelihood + priors + penalties + Jacobian) against
    gwi_nuts_run on the Python statement of the same target (sampling.make_target): same seed, same chain."""
    from gwinferno_amd.sampling import make_target, nuts_engine, nuts_native

    engs, total, prior, bij, theta0, _ = _native_nuts_setup(1)
    kw = dict(n_warmup=40, n_samples=30, seed=11, max_tree_depth=5)
    (a,) = nuts_engine(engs, total, prior, bij, [theta0], min_neff_cut=False, **kw)
    b = nuts_native(make_target(engs[0], total, prior, bij, min_neff_cut=False), bij.inverse(theta0), **kw)
    th_b = np.array([bij.forward(u)[0] for u in b["samples"]])
    assert a["n_evals"] == b["n_evals"] and np.array_equal(a["tree_depth"], b["tree_depth"])
    assert np.allclose(a["samples"], th_b, rtol=1e-8, atol=1e-10)
    assert np.allclose(a["log_prob"], b["log_prob"], rtol=1e-9)
    assert np.all(np.isfinite(a["samples"])) and 0.4 < a["accept_rate"] <= 1.0
    k = int(np.argmax(bij.kind == 1))
    assert np.all((a["samples"][:, k] > bij.lo[k]) & (a["samples"][:, k] < bij.hi[k]))
    assert np.all(a["samples"][:, int(np.argmax(bij.kind == 2))] > 0)
    engs[0].close()


def test_native_nuts_chains_in_threads_equal_chains_run_alone():
    """Four chains, one host thread and one engine each, concurrently on one GPU: every chain equals the chain
    with the same seed run alone (no cross-talk between engines, streams or host buffers)."""
    from gwinferno_amd.sampling import nuts_engine

    engs, total, prior, bij, theta0, _ = _native_nuts_setup(4)
    starts = np.stack([theta0 + 0.05 * c for c in range(4)])
    kw = dict(n_warmup=40, n_samples=30, seed=5, max_tree_depth=5)
    res = nuts_engine(engs, total, prior, bij, starts, min_neff_cut=False, **kw)
    for c, r in enumerate(res):
        kw1 = dict(kw, seed=5 + 1000 * c)
        (alone,) = nuts_engine(engs[:1], total, prior, bij, starts[c : c + 1], min_neff_cut=False, **kw1)
        assert np.array_equal(r["samples"], alone["samples"]) and r["n_evals"] == alone["n_evals"]
        assert np.std(r["samples"], axis=0).min() > 0
    with pytest.raises(ValueError):
        nuts_engine(engs[:2], total, prior, bij, starts, min_neff_cut=False, **kw)
    for e in engs:
        e.close()


def test_lockstep_chains_on_the_engine_match_the_batched_callback_path():
    """gwi_nuts_engine_lockstep (two groups of five chains, each group's leapfrog steps one gwi_eval_batch_begin / _end on
    its engine, target assembled in C++) against gwi_nuts_run_lockstep on the Python statement of the same target fed by
    the same engine's blocking gwi_eval_batch: same seeds, same batches, same chains."""
    from gwinferno_amd.sampling import nuts_engine, nuts_engine_lockstep, nuts_native_lockstep

    engs, total, prior, bij, theta0, _ = _native_nuts_setup(2)
    K = 5
    starts = np.stack([theta0 + 0.03 * c for c in range(2 * K)])
    # (a short warm-up: the two statements of the prior differ in their last bits and the Hamiltonian dynamics amplify that -- to
    # 1e-11 ... 1e-6 within sixty iterations depending on the chain, further with the large steps that follow a mass-matrix update:
    # tools/lockstep_diverge.py.  What must be EQUAL is the structure: every chain's trees, evaluation for evaluation.)
    kw = dict(n_warmup=8, n_samples=50, seed=11, max_tree_depth=5)
    res = nuts_engine_lockstep(engs, K, total, prior, bij, starts, min_neff_cut=False, **kw)
    assert len(res) == 2 * K
    again = nuts_engine_lockstep(engs, K, total, prior, bij, starts, min_neff_cut=False, **kw)
    assert all(np.array_equal(a["samples"], b["samples"]) for a, b in zip(res, again))  # deterministic
    sizes = []

    def batch_target(us, ids):
        sizes.append(len(ids))
        fw = [bij.forward(u) for u in us]
        thetas = np.stack([f[0] for f in fw])
        out = engs[0].evaluate_batch(thetas, total, min_neff_cut=False)
        lps, grads = [], []
        for (theta, dth, dlogj, logj), r in zip(fw, out):
            lp, gp = prior(theta)
            lps.append(r.log_likelihood + lp + logj)
            grads.append((r.grad + gp) * dth + dlogj)
        return np.array(lps), np.stack(grads)

    for g in range(2):
        u0 = np.stack([bij.inverse(t) for t in starts[g * K : (g + 1) * K]])
        ref = nuts_native_lockstep(batch_target, u0, **dict(kw, seed=11 + 1000 * g * K))
        for j in range(K):
            a, b = res[g * K + j], ref[j]
            th_b = np.array([bij.forward(u)[0] for u in b["samples"]])
            assert a["n_evals"] == b["n_evals"] and np.array_equal(a["tree_depth"], b["tree_depth"])
            assert np.allclose(a["samples"], th_b, rtol=1e-4, atol=1e-5)
            assert np.allclose(a["log_prob"], b["log_prob"], rtol=1e-5)
    assert max(sizes) == K and min(sizes) < K
    # lock-step chains are chains: same posterior as the threaded sampler's (pooled means within a fraction of the posterior width)
    kw = dict(n_warmup=60, n_samples=60, seed=11, max_tree_depth=5)
    res = nuts_engine_lockstep(engs, K, total, prior, bij, starts, min_neff_cut=False, **kw)
    draws = np.concatenate([r["samples"] for r in res])
    thr = nuts_engine(engs, total, prior, bij, starts[:2], min_neff_cut=False, **dict(kw, n_samples=300))
    pooled = np.concatenate([r["samples"] for r in thr])
    sd = pooled.std(axis=0)
    assert np.all(np.abs(draws.mean(axis=0) - pooled.mean(axis=0)) < 0.6 * sd)
    assert np.all(np.isfinite(draws)) and all(0.4 < r["accept_rate"] <= 1.0 for r in res)
    # more chains per group than the engine's largest batch: refused with the engine's message
    with pytest.raises(Exception, match="max_batch"):
        nuts_engine_lockstep(engs[:1], 40, total, prior, bij, np.stack([theta0] * 40), min_neff_cut=False, n_warmup=2, n_samples=2)
    one = engs[0].evaluate(theta0, total, min_neff_cut=False)  # the handle is left usable (nothing pending)
    assert np.isfinite(one.log_likelihood)
    for e in engs:
        e.close()


def test_a_queue_of_chains_on_the_engines():
    """gwi_nuts_engine_queue: eleven chains over two engines x three lock-step slots.  A chain that has drawn its last sample
    hands its slot to the next one waiting (in either group), so the batches stay full until the queue is empty; a chain's draws
    do not depend on where it ran -- checked against the callback form of the same queue (gwi_nuts_run_queue on the Python
    statement of the target, fed by the same engine's blocking batches): same trees, evaluation for evaluation."""
    from gwinferno_amd.sampling import lockstep_stats, nuts_engine_lockstep, nuts_native_lockstep

    engs, total, prior, bij, theta0, _ = _native_nuts_setup(2)
    slots, C = 3, 11
    starts = np.stack([theta0 + 0.02 * c for c in range(C)])
    kw = dict(n_warmup=8, n_samples=30, seed=5, max_tree_depth=5)
    res = nuts_engine_lockstep(engs, slots, total, prior, bij, starts, min_neff_cut=False, **kw)
    st = lockstep_stats()
    assert len(res) == C and 2.0 < st["mean_points_per_batch"] <= slots
    again = nuts_engine_lockstep(engs, slots, total, prior, bij, starts, min_neff_cut=False, **kw)
    assert all(np.array_equal(a["samples"], b["samples"]) and a["n_evals"] == b["n_evals"] for a, b in zip(res, again))  # deterministic, wherever a chain ran

    def batch_target(us, ids):
        fw = [bij.forward(u) for u in us]
        out = engs[0].evaluate_batch(np.stack([f[0] for f in fw]), total, min_neff_cut=False)
        lps, grads = [], []
        for (theta, dth, dlogj, logj), r in zip(fw, out):
            lp, gp = prior(theta)
            lps.append(r.log_likelihood + lp + logj)
            grads.append((r.grad + gp) * dth + dlogj)
        return np.array(lps), np.stack(grads)

    ref = nuts_native_lockstep(batch_target, np.stack([bij.inverse(t) for t in starts]), slots=slots, **kw)
    for a, b in zip(res, ref):  # (batches of at most three points take the same kernels and the same host-side sums in both forms)
        assert a["n_evals"] == b["n_evals"] and np.array_equal(a["tree_depth"], b["tree_depth"])
        th_b = np.array([bij.forward(u)[0] for u in b["samples"]])
        assert np.allclose(a["samples"], th_b, rtol=1e-4, atol=1e-5)
    assert np.isfinite(engs[0].evaluate(theta0, total, min_neff_cut=False).log_likelihood)  # nothing left pending
    for e in engs:
        e.close()


@pytest.mark.parametrize("name", ["chm_powerlaw", "chm_bspline"])
def test_construct_hierarchical_model_matches_the_reference(name):
    """construct_hierarchical_model (analysis.py:359-424) with the distribution classes of this package, fed the way
    the reference's function is (PopModel / PopPrior dictionaries; sampled hyper-parameters through the sample
    sites): every site the reference's own function registered for the same dictionaries (golden case_chm_*)."""
    from gwinferno_amd import interpolation as I
    from gwinferno_amd import likelihood as L
    from gwinferno_amd import numpyro_distributions as D
    from gwinferno_amd.cosmology import planck15_lvk
    from gwinferno_amd.parser import PopModel, PopPrior

    case = GoldenCase(name)
    cosmo = planck15_lvk()
    tables = {}

    def redshift(lamb, maximum, grid):  # the adapter PowerlawRedshift(zgrid=, dVcdz=) needs under this function (analysis.py:391-393)
        dv = tables.setdefault(id(grid), (grid, cosmo.dVc_dz(grid)))[1]
        return D.PowerlawRedshift(lamb, maximum, zgrid=grid, dVcdz=dv)

    if name == "chm_powerlaw":
        model_dict = {"mass_1": PopModel(D.Powerlaw, ["alpha", "minimum", "maximum"]), "mass_ratio": PopModel(D.Powerlaw, ["alpha", "minimum", "maximum"]),
                      "redshift": PopModel(redshift, ["lamb", "maximum"])}
        sampled = lambda p: {"mass_1_alpha": p["alpha"], "mass_1_minimum": p["mmin"], "mass_1_maximum": p["mmax"], "mass_ratio_alpha": p["beta"], "redshift_lamb": p["lamb"]}  # noqa: E731
        consts = dict(mass_ratio_minimum=0.02, mass_ratio_maximum=1.0, redshift_maximum=1.9)
    else:
        m_grid, q_grid = np.linspace(case.meta["mmin"], case.meta["mmax"], 1000), np.linspace(0.0, 1.0, 1000)
        m_dmat = I.LogXLogYBSpline(16, xrange=(case.meta["mmin"], case.meta["mmax"]), normalize=True).bases(m_grid)
        q_dmat = I.LogYBSpline(10, xrange=(0.0, 1.0), normalize=True).bases(q_grid)
        names = ["minimum", "maximum", "cs", "grid", "grid_dmat"]
        model_dict = {"mass_1": PopModel(D.BSplineDistribution, names), "mass_ratio": PopModel(D.BSplineDistribution, names), "redshift": PopModel(redshift, ["lamb", "maximum"])}
        sampled = lambda p: {"mass_1_cs": p["m_coefs"], "mass_ratio_cs": p["q_coefs"], "redshift_lamb": p["lamb"]}  # noqa: E731
        consts = dict(mass_1_minimum=case.meta["mmin"], mass_1_maximum=case.meta["mmax"], mass_1_grid=m_grid, mass_1_grid_dmat=m_dmat, mass_ratio_minimum=0.0, mass_ratio_maximum=1.0,
                      mass_ratio_grid=q_grid, mass_ratio_grid_dmat=q_dmat, redshift_maximum=1.9)
    prior_dict = {k: PopPrior(None, {}) for k in sampled(case.point(0))}
    prior_dict.update(consts)
    L.SAMPLE_VALUES["unscaled_rate"] = case.meta["unscaled_rate"]
    L.clear_engine_cache()
    for fs, flags in case.flagsets.items():
        flags = {k: v for k, v in flags.items() if k != "log"}
        model = L.construct_hierarchical_model(model_dict, prior_dict, posterior_predictive_check=False, **flags)
        for i in range(case.n_points):
            L.SAMPLE_VALUES.update(sampled(case.point(i)))
            model(case.pe, case.inj, case.total_inj, case.nobs, case.tobs)
            sites = L.last_sites()
            for site, ref in case.sites[fs].items():
                if site.startswith("variance"):
                    assert np.allclose(sites[site], ref[i], rtol=1e-8, atol=1e-12), (fs, i, site)
                else:
                    assert rel_err(sites[site], ref[i]) < 1e-9, (fs, i, site, sites[site], ref[i])
            if not flags.get("marginalize_selection"):
                assert np.all(np.isfinite(sites["grad_log_likelihood"]))
    # one engine per constructed model (each construct call makes its own redshift grid, analysis.py:371-372), however
    # many times the model function ran: nothing is keyed on the per-call distribution objects
    assert len(L._ENGINES) == len(case.flagsets)
    # the reference's DEFAULT call, construct_hierarchical_model(model_dict, prior_dict): posterior_predictive_check=True
    # (analysis.py:359-366) -- runs, gives the same likelihood sites, and adds one observed / predicted draw per event
    # and source parameter (:350-355), taken from the catalog with the per-sample weights of the engine
    L.construct_hierarchical_model(model_dict, prior_dict)(case.pe, case.inj, case.total_inj, case.nobs, case.tobs)
    sites = L.last_sites()
    assert rel_err(sites["log_l"], case.sites["log_neff"]["log_l"][case.n_points - 1]) < 1e-9  # defaults == the "log_neff" flag set, last hyper-point
    eng, pe_w, _, _ = L._ENGINES[next(reversed(L._ENGINES))]
    lw_pe, _ = eng.log_weights(eng.bound.theta_of(pe_w))
    for ev in range(case.nobs):
        for p in model_dict:
            obs, pred = sites[f"{p}_obs_event_{ev}"], sites[f"{p}_pred_event_{ev}"]
            j = np.nonzero(case.pe[p][ev] == obs)[0]
            assert j.size >= 1 and pred in case.inj[p]
        j_obs = int(np.nonzero(case.pe["mass_1"][ev] == sites[f"mass_1_obs_event_{ev}"])[0][0])
        assert np.isfinite(lw_pe[ev, j_obs])  # a drawn sample carries weight
        m1, q = case.pe["mass_1"][ev, j_obs], case.pe["mass_ratio"][ev, j_obs]
        assert 2.0 <= m1 <= 100.0 and m1 * q >= 2.0  # the mass cuts construct_hierarchical_model passes (analysis.py:417-419)
    L.clear_engine_cache()


def test_pipeline_factories_and_example_priors_end_to_end(monkeypatch):
    """The reference's B-spline example (examples/simple_bspline_example.py:26-94) through this package: models from
    the pipeline_utils factories (weights == the reference's, tests/golden/pipeline.npz), its priors handed to the
    library's sampler (bspline_example_prior), the first redshift coefficient pinned to 0 through a FIXED slot.
    The engine runs in replay mode: the test compares two NUTS chains sample by sample, and the last-bit run-to-run
    differences of spline-coefficient gradients in the regular mode (LDS atomics from four wavefronts) are enough to move a
    chaotic trajectory past 1e-8 now and then (2 runs in 10)."""
    monkeypatch.setenv("GWI_DETERMINISTIC", "1")
    import json
    import os

    from golden_util import GOLDEN_DIR
    from test_pipeline_utils_cpu import _example_product

    from gwinferno_amd import pipeline_utils as U
    from gwinferno_amd.engine import NativePopulationLikelihood
    from gwinferno_amd.sampling import make_target, nuts_engine, nuts_native

    fx = np.load(os.path.join(GOLDEN_DIR, "pipeline.npz"))
    wp, wi, hv = _example_product(fx)
    eng = NativePopulationLikelihood(wp, wi, hv)
    theta = eng.bound.theta_of(wp)
    lpe, linj = eng.log_weights(theta)
    for got, ref in ((lpe, fx["factory/pe"]), (linj, fx["factory/inj"])):
        with np.errstate(all="ignore"):
            rl = np.log(ref)
        dead = ~(ref > 0)
        assert np.array_equal(np.isneginf(got), dead) and np.max(np.abs(got[~dead] - rl[~dead])) < 1e-10
    # theta layout: find each coefficient block by value
    cs = {k[7:]: fx[k] for k in fx.files if k.startswith("sample/")}
    z_full = np.concatenate([np.zeros(1), cs["z_cs"]])

    def where(block):
        for off in range(eng.n_theta - len(block) + 1):
            if np.array_equal(theta[off : off + len(block)], block):
                return slice(off, off + len(block))
        raise AssertionError("block not found")

    slices = {"m1": where(cs["mass_cs"]), "q": where(cs["q_cs"]), "a1": where(cs["a1_cs"]), "a2": where(cs["a2_cs"]), "tilt1": where(cs["tilt1_cs"]), "tilt2": where(cs["tilt2_cs"]),
              "redshift": where(z_full), "lamb": where(np.array([float(fx["lamb"])]))}
    assert sum(s.stop - s.start for s in slices.values()) == eng.n_theta
    prior, bij = U.bspline_example_prior(slices)
    total = 20.0 * json.loads(str(fx["meta"]))["catalog"][2]
    start = theta * 0.2  # a mild point: short trees
    # 40 warm-up iterations: the windowed schedule (6 / 30 / 4) leaves the step size four iterations after its restart at the
    # end of the metric window; with 25 (3 / 20 / 2) it stays near the restart's 10 x initial guess and every transition
    # diverges -- Stan's and NumPyro's schedule does the same with so short a warm-up
    kw = dict(n_warmup=40, n_samples=15, seed=2, max_tree_depth=3)
    (a,) = nuts_engine([eng], total, prior, bij, [start], min_neff_cut=False, **kw)
    zi = slices["redshift"].start
    assert np.all(a["samples"][:, zi] == 0.0) and np.std(np.delete(a["samples"], zi, axis=1), axis=0).min() > 0
    b = nuts_native(make_target(eng, total, prior, bij, min_neff_cut=False), bij.inverse(start), **kw)
    th_b = np.array([bij.forward(u)[0] for u in b["samples"]])
    # the two samplers evaluate priors and bijectors with independent arithmetic (C++ / NumPy): last-bit differences there,
    # amplified along the ~ 385 leapfrog steps of the run, reach a few 1e-10 (measured 3.6e-10 .. 5.8e-10 over 576 steps); the tree structure is identical
    assert a["n_evals"] == b["n_evals"] and np.allclose(a["samples"], th_b, rtol=1e-7, atol=1e-8)
    eng.close()


def test_torch_autograd_adapter():
    """torch_adapter.log_likelihood: value and theta.grad equal the engine's; it composes with torch ops (a prior term,
    a reparametrisation) and a few L-BFGS steps climb the log-posterior."""
    import torch

    from gwinferno_amd.compositions import COMPOSITIONS
    from gwinferno_amd.synthetic import make_catalog
    from gwinferno_amd.torch_adapter import log_likelihood

    pe, inj, total = make_catalog(20, 400, 4000, seed=3)
    comp = COMPOSITIONS["pl_test"](pe, inj)
    eng = comp.engine()
    th0 = comp.theta({"alpha": -2.0, "beta": 1.0, "lamb": 2.0})
    ref = eng.evaluate(th0, total, min_neff_cut=False)
    theta = torch.tensor(th0, dtype=torch.float64, requires_grad=True)
    ll = log_likelihood(eng, theta, total, min_neff_cut=False)
    ll.backward()
    assert float(ll.detach()) == ref.log_likelihood and np.array_equal(theta.grad.numpy(), ref.grad)
    # chain rule through torch: theta = 2 u, objective = ll - 0.5 |u|^2
    u = torch.tensor(th0 / 2.0, dtype=torch.float64, requires_grad=True)
    obj = log_likelihood(eng, 2.0 * u, total, min_neff_cut=False) - 0.5 * (u**2).sum()
    (g,) = torch.autograd.grad(obj, u)
    assert np.allclose(g.numpy(), 2.0 * ref.grad - th0 / 2.0, rtol=1e-14, atol=1e-14)
    x = torch.tensor(th0, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.LBFGS([x], lr=0.5, max_iter=15, line_search_fn="strong_wolfe")

    def closure():
        opt.zero_grad()
        loss = -(log_likelihood(eng, x, total, min_neff_cut=False) - 0.5 * (x**2).sum() / 25.0)
        loss.backward()
        return loss

    first = float(closure().detach())
    opt.step(closure)
    assert float(closure().detach()) < first - 1e-3 and torch.all(torch.isfinite(x))
    with pytest.raises(ValueError):
        log_likelihood(eng, torch.zeros(2, dtype=torch.float64), total)
    eng.close()


def test_engine_cache_is_bounded_and_keyed_on_live_objects():
    """A model function that hands in NEW arrays on every call gets a new engine every call (the key is the identity of
    the data): the cache evicts (and closes) the least recently used engine instead of filling the HBM, and keeps the
    keyed arrays alive so that a recycled id can never select an engine holding other data."""
    import gc

    from gwinferno_amd import likelihood as L
    from gwinferno_amd.lazy import where_finite
    from gwinferno_amd.models import PowerlawRedshiftModel, powerlaw_primary_ratio_pdf
    from gwinferno_amd.synthetic import make_catalog

    L.clear_engine_cache()
    L.SAMPLE_VALUES["unscaled_rate"] = 30.0
    values = {}
    for rep in range(2):
        for seed in range(11):
            pe, inj, total = make_catalog(4, 64, 600, seed=100 + seed)  # fresh arrays every time
            z_model = PowerlawRedshiftModel(z_pe=pe["redshift"], z_inj=inj["redshift"])
            w = lambda d: where_finite(powerlaw_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], alpha=-2.2, beta=1.0, mmin=5.0, mmax=100.0) * z_model(d["redshift"], 2.0) / d["prior"])  # noqa: E731
            L.hierarchical_likelihood(w(pe), w(inj), total_inj=total, Nobs=4, Tobs=1.0, surveyed_hypervolume=z_model.normalization(2.0), min_neff_cut=False)
            ll = L.last_sites()["log_likelihood"]
            assert values.setdefault(seed, ll) == ll  # same data -> same value, whichever engine served it
            del pe, inj, z_model, w
            gc.collect()
            assert len(L._ENGINES) <= 8
    assert len(set(values.values())) == 11
    L.clear_engine_cache()


def test_jax_host_callback_routes_batches_through_the_batched_kernels():
    """The host side of the JAX seam with a batch of points (what ``vmap`` of the model hands ``pure_callback`` under
    ``vmap_method="broadcast_all"``): 20 points of the config-3 model go through ``gwi_eval_batch`` (two launch sets) and agree
    with 20 single evaluations."""
    from gwinferno_amd import likelihood as L
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(12, 400, 6000, seed=31)
    comp = COMPOSITIONS["bspline_iid"](pe, inj)
    eng = comp.engine()
    flags = dict(marginalize_selection=False, min_neff_cut=False, max_variance_cut=False)
    host = L._host_callback(eng, total, 12, flags)
    rng = np.random.default_rng(2)
    thetas = np.stack([eng.bound.theta_of(comp.weights(draw_params("bspline_iid", rng), True)) for _ in range(20)])
    summ, per_event, grad = host(thetas)
    assert summ.shape == (20, len(L._SUMMARY_FIELDS)) and per_event.shape == (20, 3, 12) and grad.shape == (20, eng.n_theta)
    assert eng.lib.gwi_batch_path(eng.handle, 16).decode() in ("mfma", "taps", "rows", "pbatch", "rows-per-point")
    for k in (0, 7, 19):
        s1, p1, g1 = host(thetas[k])
        assert abs(s1[0] - summ[k, 0]) <= 1e-11 * abs(s1[0]) and np.allclose(p1, per_event[k], rtol=1e-10, atol=1e-10)
        assert np.allclose(g1, grad[k], rtol=1e-9, atol=1e-9 * np.max(np.abs(g1)))
    eng.close()


@pytest.mark.parametrize("log", [False, True])
def test_array_valued_weights_through_the_engine(log):
    """The reference's reductions take plain arrays of weights (analysis.py:50-163).  The drop-ins accept them too: the same
    scan kernel on exp(kappa) x a unit factor.  Golden: the unmodified reference functions on seeded arrays spanning 40
    e-folds with exact zeros (tests/golden/make_golden.py arrays), linear and log."""
    import os

    from gwinferno_amd import likelihood as L

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "array_weights.npz"))
    with np.errstate(all="ignore"):
        a, b = (g["lw_pe"], g["lw_inj"]) if log else (np.exp(g["lw_pe"]), np.exp(g["lw_inj"]))
    tag = "log" if log else "lin"
    total = float(g["total_inj"])
    lbf, lne, var = L.per_event_log_bayes_factors(a, log=log)
    assert rel_err(lbf, g[f"{tag}/pe/logBFs"]) < 1e-9 and rel_err(lne, g[f"{tag}/pe/log_nEffs"]) < 1e-9 and rel_err(var, g[f"{tag}/pe/variances"]) < 1e-8
    lmu, lneff, v = L.detection_efficiency(b, total, log=log)
    assert rel_err(lmu, g[f"{tag}/inj/logmu"]) < 1e-9 and rel_err(lneff, g[f"{tag}/inj/log_nEff"]) < 1e-9 and rel_err(v, g[f"{tag}/inj/variance"]) < 1e-8
    with pytest.raises(ValueError):
        L.detection_efficiency(a, total, log=log)  # a (N_ev, N_pe) array is not an injection set
    L.SAMPLE_VALUES["unscaled_rate"] = 30.0
    for fname, flags in (("cut", dict(min_neff_cut=True)), ("nocut", dict(min_neff_cut=False)), ("marg", dict(min_neff_cut=False, marginalize_selection=True))):
        rate = L.hierarchical_likelihood(a, b, total_inj=total, Nobs=a.shape[0], Tobs=1.0, surveyed_hypervolume=float(g["hypervolume"]), log=log, **flags)
        sites = L.last_sites()
        for k in ("log_likelihood", "log_l", "logBFs", "log_nEffs", "log_nEff_inj", "detection_efficiency", "selection_factor", "sum_logBFs", "variance_log_BFs",
                  "variance_log_detection_efficiency", "variance_log_likelihood", "surveyed_hypervolume", "rate"):
            want = g[f"{tag}/hl_{fname}/{k}"]
            assert np.array_equal(np.isfinite(sites[k]), np.isfinite(want)), (fname, k)
            assert rel_err(sites[k], want) < (1e-8 if "variance" in k else 1e-9), (fname, k)
        assert rel_err(rate, g[f"{tag}/hl_{fname}/rate_return"]) < 1e-9
    with pytest.raises(TypeError):  # arrays carry no normaliser: the hypervolume is the caller's number
        L.hierarchical_likelihood(a, b, total_inj=total, Nobs=a.shape[0], Tobs=1.0, log=log)
    L.clear_engine_cache()
