"""CPU: the NumPyro seam as far as an image without JAX allows.  ``likelihood._evaluate_jax`` -- the ``jax.custom_vjp``
around ``jax.pure_callback`` that lets ``jit(value_and_grad(potential_energy))`` (tests/inference_test.py:320-326,
pipeline/analysis.py:260-319 of the reference) drive the engine unchanged -- runs here under a minimal ``jax`` shim
(tests/jaxshim: custom_vjp, pure_callback, ShapeDtypeStruct, a few jax.numpy functions over an opaque tracer).  The
engine behind the callback is the C oracle (this tests adapter plumbing, not kernels): shapes declared to
pure_callback, forward rule == primal, the VJP returns ct * grad, every site of the reference survives as a traced
value, and the posterior-predictive branch (analysis.py:321-355) produces its sites through a second callback."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "jaxshim")


class _OracleEngine:
    """Stand-in for NativePopulationLikelihood.evaluate / .log_weights on a host-only handle."""

    def __init__(self, eng):
        from oracle.c_oracle import COracle

        self._eng, self._orc = eng, COracle(eng.bound)
        self.calls = 0

    def __getattr__(self, name):
        return getattr(self._eng, name)

    def evaluate(self, theta, total_inj, nobs=None, marginalize_selection=False, min_neff_cut=True, max_variance_cut=False, want_grad=True, copy=True):
        from gwinferno_amd.engine import EvalResult

        self.calls += 1
        r = self._orc.evaluate(theta, total_inj, nobs=nobs, marginalize_selection=marginalize_selection, min_neff_cut=min_neff_cut, max_variance_cut=max_variance_cut)
        return EvalResult(log_likelihood=r["log_likelihood"], grad=r["grad"] if want_grad else None, summary=r["summary"], log_bfs=r["logBFs"], log_neffs=r["log_nEffs"],
                          variances=r["variance_log_BFs"], norms=r["norms"])

    def evaluate_batch(self, thetas, total_inj, **kw):
        return [self.evaluate(t, total_inj, **kw) for t in np.asarray(thetas)]

    def log_weights(self, theta):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from bound_eval import log_weights

        lpe, linj, _ = log_weights(self._eng.bound, theta, include_consts=True)
        return lpe, linj


@pytest.fixture
def shim(monkeypatch):
    monkeypatch.syspath_prepend(SHIM)
    for m in [k for k in sys.modules if k == "jax" or k.startswith("jax.")]:
        monkeypatch.delitem(sys.modules, m)
    import jax

    assert jax.__file__.startswith(SHIM)
    yield jax
    for m in [k for k in sys.modules if k == "jax" or k.startswith("jax.")]:
        sys.modules.pop(m, None)


def _model_pieces():
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(6, 64, 600, seed=17)
    return pe, inj, total


def test_custom_vjp_adapter_under_the_shim(shim, monkeypatch):
    jax = shim
    import jax.numpy as jnp

    from gwinferno_amd import _native as N
    from gwinferno_amd import likelihood as L
    from gwinferno_amd.engine import NativePopulationLikelihood
    from gwinferno_amd.lazy import where_finite
    from gwinferno_amd.models import PowerlawRedshiftModel, powerlaw_primary_ratio_pdf

    pe, inj, total = _model_pieces()
    z_model = PowerlawRedshiftModel(z_pe=pe["redshift"], z_inj=inj["redshift"])
    made = {}

    def fake_engine_for(pe_w, inj_w, hv=None, device=-1):
        if "eng" not in made:
            made["eng"] = _OracleEngine(NativePopulationLikelihood(pe_w, inj_w, hv, device=N.DEVICE_HOST_ONLY))
        return made["eng"]

    monkeypatch.setattr(L, "engine_for", fake_engine_for)
    monkeypatch.setattr(L, "_NUMPYRO", [None])
    L.SAMPLE_VALUES["unscaled_rate"] = 6.0

    def model(alpha, beta, lamb, **kw):  # tests/inference_test.py:162-197 with the drop-in names
        def get_weights(d):
            return where_finite(powerlaw_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], alpha=alpha, beta=beta, mmin=5.0, mmax=100.0) * z_model(d["redshift"], lamb) / d["prior"])

        return L.hierarchical_likelihood(get_weights(pe), get_weights(inj), total_inj=total, Nobs=6, Tobs=1.0, surveyed_hypervolume=z_model.normalization(lamb=lamb),
                                         min_neff_cut=False, **kw)

    point = dict(alpha=-2.3, beta=0.8, lamb=2.5)
    # concrete numbers first: the NumPy branch is the yardstick
    rate_np = model(**point)
    want = L.last_sites()
    # ... then the same call with traced hyper-parameters (what NUTS hands the model function under jit)
    traced = {k: jnp.asarray(v) for k, v in point.items()}
    n_before = len(jax.CALLBACK_CALLS)
    rate_tr = model(**traced)
    got = L.last_sites()
    assert len(jax.CALLBACK_CALLS) == n_before + 2  # primal + forward rule, each one pure_callback
    assert isinstance(got["log_likelihood"], jnp.Tracer) and isinstance(rate_tr, jnp.Tracer)
    for name, ref in want.items():
        if name == "grad_log_likelihood":
            continue  # the traced call hands the gradient to the VJP, not to a site
        val = got[name]
        assert isinstance(val, jnp.Tracer), name
        assert np.allclose(val.val, ref, rtol=1e-14, atol=0), name
    assert np.allclose(rate_tr.val, rate_np, rtol=1e-14)
    # the backward rule: cotangent on log_likelihood (summary[0]) -> ct * d log_l / d theta; nothing else is differentiable
    n_sum = len(L._SUMMARY_FIELDS)
    ct_summ = np.zeros(n_sum)
    ct_summ[0] = 3.0
    ct_summ[4] = 100.0  # a cotangent on a diagnostic (log_det_eff) must not leak into theta
    (g,) = jax.custom_vjp.pull((ct_summ, np.ones((3, 6))))
    assert g.shape == (made["eng"].n_theta,)
    assert np.allclose(g, 3.0 * want["grad_log_likelihood"], rtol=1e-14, atol=0)
    # theta reaches the host function in layout order
    layout_theta = made["eng"].bound.theta_of(powerlaw_primary_ratio_pdf(pe["mass_1"], pe["mass_ratio"], alpha=-2.3, beta=0.8, mmin=5.0, mmax=100.0) * z_model(pe["redshift"], 2.5) / pe["prior"])
    assert sorted(layout_theta) == sorted(point.values())


def test_posterior_predictive_sites_numpy_and_traced(shim, monkeypatch):
    """analysis.py:320-355 -- `posterior_predictive_check=True`, the reference's default in construct_hierarchical_model:
    one (obs, pred) index pair per event drawn from the masked weights; identical sites from the NumPy and the traced call."""
    import jax.numpy as jnp

    from gwinferno_amd import _native as N
    from gwinferno_amd import likelihood as L
    from gwinferno_amd.engine import NativePopulationLikelihood
    from gwinferno_amd.lazy import where_finite
    from gwinferno_amd.models import PowerlawRedshiftModel, powerlaw_primary_ratio_pdf

    pe, inj, total = _model_pieces()
    z_model = PowerlawRedshiftModel(z_pe=pe["redshift"], z_inj=inj["redshift"])
    made = {}

    def fake_engine_for(pe_w, inj_w, hv=None, device=-1):
        if "eng" not in made:
            made["eng"] = _OracleEngine(NativePopulationLikelihood(pe_w, inj_w, hv, device=N.DEVICE_HOST_ONLY))
        return made["eng"]

    monkeypatch.setattr(L, "engine_for", fake_engine_for)
    monkeypatch.setattr(L, "_NUMPYRO", [None])
    names = ["mass_1", "mass_ratio", "redshift"]

    def model(alpha, beta, lamb):
        def get_weights(d):
            return where_finite(powerlaw_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], alpha=alpha, beta=beta, mmin=5.0, mmax=100.0) * z_model(d["redshift"], lamb) / d["prior"])

        return L.hierarchical_likelihood(get_weights(pe), get_weights(inj), total_inj=total, Nobs=6, Tobs=1.0, surveyed_hypervolume=z_model.normalization(lamb=lamb),
                                         min_neff_cut=False, posterior_predictive_check=True, param_names=names, pedata=pe, injdata=inj, m1min=5.0, m2min=3.0, mmax=100.0)

    model(alpha=-2.3, beta=0.8, lamb=2.5)
    a = L.last_sites()
    model(alpha=jnp.asarray(-2.3), beta=jnp.asarray(0.8), lamb=jnp.asarray(2.5))
    b = L.last_sites()
    lw_pe, lw_inj = made["eng"].log_weights(np.array([-2.3, 0.8, 2.5]))
    for ev in range(6):
        for p in names:
            obs, pred = a[f"{p}_obs_event_{ev}"], a[f"{p}_pred_event_{ev}"]
            assert obs == b[f"{p}_obs_event_{ev}"].val and pred == b[f"{p}_pred_event_{ev}"].val
            assert obs in pe[p][ev] and pred in inj[p]
        # a drawn sample has non-zero weight and passes the reference's mass cuts (:325-337)
        j = int(np.nonzero(pe["mass_1"][ev] == a[f"mass_1_obs_event_{ev}"])[0][0])
        assert np.isfinite(lw_pe[ev, j]) and 5.0 <= pe["mass_1"][ev, j] <= 100.0 and pe["mass_1"][ev, j] * pe["mass_ratio"][ev, j] >= 3.0
    # over many events' worth of draws the index distribution follows the weights: chi^2-free sanity check on event 0
    w = np.exp(lw_pe[0] - lw_pe[0].max())
    w[(pe["mass_1"][0] < 5.0) | (pe["mass_1"][0] > 100.0) | (pe["mass_1"][0] * pe["mass_ratio"][0] < 3.0)] = 0.0
    cdf = np.cumsum(w)
    u = np.random.default_rng([0, 0]).uniform() * cdf[-1]
    assert pe["mass_1"][0, min(int(np.searchsorted(cdf, u, side="right")), w.size - 1)] == a["mass_1_obs_event_0"]


def test_evicted_engines_stay_usable(monkeypatch):
    """cache eviction must not destroy a handle somebody still holds (the jitted NUTS step captures it)."""
    from gwinferno_amd import likelihood as L

    class Fake:
        closed = False

        def close(self):
            self.closed = True

    made = []

    def fake_ctor(*a, **k):
        made.append(Fake())
        return made[-1]

    monkeypatch.setattr(L, "NativePopulationLikelihood", fake_ctor)
    monkeypatch.setattr(L, "structure_key", lambda a, b: (a, b))
    monkeypatch.setenv("GWI_ENGINE_CACHE", "2")
    L._ENGINES.clear()
    held = [L.engine_for(i, i) for i in range(5)]
    assert len(L._ENGINES) == 2 and not any(e.closed for e in held)
    L.clear_engine_cache()
    assert not any(e.closed for e in held)


def test_host_callback_takes_batches_of_points():
    """What a vmapped model (NumPyro's vectorised chains) hands the callback under ``vmap_method="broadcast_all"``: theta with
    leading batch dimensions.  The host side answers with the same leading dimensions on every output, point for point what
    single calls give."""
    from gwinferno_amd import _native as N
    from gwinferno_amd import likelihood as L
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.engine import NativePopulationLikelihood

    pe, inj, total = _model_pieces()
    comp = COMPOSITIONS["plpeak"](pe, inj)
    p = comp.placeholder()
    eng = _OracleEngine(NativePopulationLikelihood(comp.weights(p, True), comp.weights(p, False), comp.hypervolume(p), device=N.DEVICE_HOST_ONLY))
    flags = dict(marginalize_selection=False, min_neff_cut=False, max_variance_cut=False)
    host = L._host_callback(eng, total, 6, flags)
    rng = np.random.default_rng(4)
    thetas = np.stack([eng.bound.theta_of(comp.weights(draw_params("plpeak", rng), True)) for _ in range(6)])
    singles = [host(t) for t in thetas]
    summ, per_event, grad = host(thetas)                      # one batch dimension
    assert summ.shape == (6, len(L._SUMMARY_FIELDS)) and per_event.shape == (6, 3, 6) and grad.shape == (6, eng.n_theta)
    for k in range(6):
        assert all(np.array_equal(a, b) for a, b in zip(singles[k], (summ[k], per_event[k], grad[k])))
    summ2, per_event2, grad2 = host(thetas.reshape(2, 3, -1))  # nested vmaps: two
    assert summ2.shape == (2, 3, len(L._SUMMARY_FIELDS)) and np.array_equal(summ2.reshape(6, -1), summ) and np.array_equal(grad2.reshape(6, -1), grad)
    assert per_event2.shape == (2, 3, 3, 6)


def test_jax_check_command_reports_a_missing_jax(capsys):
    """``python -m gwinferno_amd.jax_check`` on a box without JAX (this image): a JSON report that says so, exit status 2."""
    import json

    pytest.importorskip("numpy")
    try:
        import jax  # noqa: F401
    except ImportError:
        pass
    else:
        pytest.skip("a real JAX is importable here")
    from gwinferno_amd import jax_check

    assert jax_check.main([]) == 2
    report = json.loads(capsys.readouterr().out)
    assert "jax / numpyro not importable" in report["error"] and "python" in report


def _spline_model(L, pe, inj, total, comp, nobs, **flags):
    """A model function with VECTOR-valued sites (the B-spline coefficient vectors of tests/inference_test.py:244-285) written
    with the drop-in names; returns the potential NUTS differentiates, U = -(log_likelihood factor + log priors), with unit
    normal priors on every coefficient (:228-229) -- built from the traced `log_likelihood` site."""
    import jax.numpy as jnp

    def potential(params):
        L.hierarchical_likelihood(comp.weights(params, True), comp.weights(params, False), total_inj=total, Nobs=nobs, Tobs=1.0,
                                  surveyed_hypervolume=comp.hypervolume(params), **flags)
        ll = L.last_sites()["log_likelihood"]
        prior = 0.0
        for name, shape in comp.PARAMS.items():
            if shape:
                prior = prior + (-0.5) * jnp.sum(params[name] * params[name])
        return -(ll + prior)

    return potential


def _spline_setup(monkeypatch):
    from gwinferno_amd import _native as N
    from gwinferno_amd import likelihood as L
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.engine import NativePopulationLikelihood

    pe, inj, total = _model_pieces()
    comp = COMPOSITIONS["bspline_test"](pe, inj)
    made = {}

    def fake_engine_for(pe_w, inj_w, hv=None, device=-1):
        if "eng" not in made:
            made["eng"] = _OracleEngine(NativePopulationLikelihood(pe_w, inj_w, hv, device=N.DEVICE_HOST_ONLY))
        return made["eng"]

    monkeypatch.setattr(L, "engine_for", fake_engine_for)
    monkeypatch.setattr(L, "_NUMPYRO", [None])
    monkeypatch.setattr(L, "_WARNED_F32", [])
    L.SAMPLE_VALUES["unscaled_rate"] = 6.0
    rng = np.random.default_rng(21)
    points = [draw_params("bspline_test", rng) for _ in range(5)]
    return L, comp, pe, inj, total, made, points


def _expected(L, comp, potential, p, made):
    """value and gradient of the potential from the NumPy branch (concrete numbers): the yardstick for the traced calls"""
    pn = {k: np.asarray(v, dtype=np.float64) for k, v in p.items()}
    # the NumPy branch wants plain numbers; the prior part is evaluated by hand
    L.hierarchical_likelihood(comp.weights(pn, True), comp.weights(pn, False), total_inj=potential.total, Nobs=potential.nobs, Tobs=1.0,
                              surveyed_hypervolume=comp.hypervolume(pn), **potential.flags)
    s = L.last_sites()
    theta = made["eng"].bound.theta_of(comp.weights(pn, True))
    value = -(s["log_likelihood"] + sum(-0.5 * float(np.sum(pn[k] ** 2)) for k, shp in comp.PARAMS.items() if shp))
    grads = {}
    for k, shp in comp.PARAMS.items():
        v = np.ravel(pn[k])
        off = next(o for o in range(len(theta) - v.size + 1) if np.array_equal(theta[o : o + v.size], v))
        g = -np.asarray(s["grad_log_likelihood"][off : off + v.size])
        grads[k] = (g + v).reshape(shp) if shp else g.reshape(())
    return value, grads


@pytest.mark.parametrize("x64", [True, False])
def test_value_and_grad_of_a_potential_with_vector_sites(shim, monkeypatch, x64):
    """jit(value_and_grad(potential)) -- what NUTS runs (tests/inference_test.py:320-326) -- over a dict of parameters that holds
    coefficient VECTORS, under the shim's reverse sweep: the custom_vjp's backward rule delivers d log_l / d theta, split back
    onto the sites in layout order, next to the prior terms JAX differentiates itself.  With jax_enable_x64 OFF (JAX's default,
    what a reference user has unless they export JAX_ENABLE_X64) everything crosses the seam as float32 and says so once."""
    jax = shim
    L, comp, pe, inj, total, made, points = _spline_setup(monkeypatch)
    jax.config.update("jax_enable_x64", x64)
    try:
        flags = dict(min_neff_cut=False)
        potential = _spline_model(L, pe, inj, total, comp, 6, **flags)
        potential.total, potential.nobs, potential.flags = total, 6, flags
        want_v, want_g = _expected(L, comp, potential, points[0], made)
        import warnings

        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            value, grads = jax.jit(jax.value_and_grad(potential))(points[0])
            jax.jit(jax.value_and_grad(potential))(points[0])
        said = [w for w in caught if "jax_enable_x64" in str(w.message)]
        assert len(said) == (0 if x64 else 1)  # once, naming the switch
        dt = np.float64 if x64 else np.float32
        tol = 1e-12 if x64 else 2e-5
        assert value.dtype == dt and abs(float(value) - want_v) <= tol * abs(want_v)
        for k, shp in comp.PARAMS.items():
            assert grads[k].dtype == dt and grads[k].shape == shp, k
            assert np.allclose(grads[k], want_g[k], rtol=tol, atol=tol * max(1.0, float(np.max(np.abs(want_g[k]))))), k
        # a cut replaces log_l by nan_to_num(-inf): finite in either precision (analysis.py:280-303), with a zero gradient
        cut = _spline_model(L, pe, inj, total, comp, 5000, min_neff_cut=True)  # 4 N_obs far above any n_eff of 600 injections
        v, g = jax.value_and_grad(cut)(points[0])
        assert np.isfinite(v) and float(v) >= 0.99 * float(np.finfo(dt).max)
        assert all(np.allclose(g[k], np.asarray(points[0][k], dtype=np.float64), rtol=1e-6) for k, shp in comp.PARAMS.items() if shp)  # only the prior part is left
    finally:
        jax.config.update("jax_enable_x64", True)


def test_vmap_over_chains_makes_one_batched_callback(shim, monkeypatch):
    """NumPyro's vectorised chains: vmap(value_and_grad(potential)) over K parameter sets.  The callback is declared with
    vmap_method="broadcast_all", so the K points reach the host in ONE call (gwi_eval_batch behind it) and every chain gets
    what a call of its own gives."""
    jax = shim
    L, comp, pe, inj, total, made, points = _spline_setup(monkeypatch)
    flags = dict(min_neff_cut=False)
    potential = _spline_model(L, pe, inj, total, comp, 6, **flags)
    potential.total, potential.nobs, potential.flags = total, 6, flags
    singles = [jax.value_and_grad(potential)(p) for p in points]
    batched = {k: np.stack([np.asarray(p[k], dtype=np.float64) for p in points]) for k in comp.PARAMS}
    n_before, calls_before = len(jax.CALLBACK_CALLS), made["eng"].calls
    values, grads = jax.vmap(jax.value_and_grad(potential))(batched)
    assert len(jax.CALLBACK_CALLS) == n_before + 2  # primal + forward rule: one batched call each, not one per chain
    assert made["eng"].calls == calls_before + 2 * len(points)
    assert values.shape == (len(points),)
    for i, (v, g) in enumerate(singles):
        assert np.allclose(values[i], v, rtol=1e-13)
        for k, shp in comp.PARAMS.items():
            assert grads[k].shape == (len(points),) + shp
            assert np.allclose(grads[k][i], g[k], rtol=1e-12, atol=1e-12), (i, k)
