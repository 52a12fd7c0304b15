"""CPU: the host-side model mirrors (gwinferno_amd.models / compositions / engine.bind) produce
columns, masks, normaliser grids and a theta layout that reproduce the reference's per-sample
weights and hierarchical_likelihood sites (golden vectors), when evaluated by the NumPy
BoundModel evaluator in tests/bound_eval.py.  The HIP kernels are checked against the same
vectors in test_gpu_parity.py (-m gpu)."""
import numpy as np
import os

import pytest
from bound_eval import log_weights
from golden_util import CASES, GoldenCase, rel_err

from gwinferno_amd import _native as N
from gwinferno_amd.compositions import COMPOSITIONS
from gwinferno_amd.engine import bind, shard_bounds
from oracle import numpy_oracle as O


def _bound(case):
    comp = COMPOSITIONS[case.composition](case.pe, case.inj, mmin=case.meta["mmin"], mmax=case.meta["mmax"])
    p = case.point(0)
    return comp, bind(comp.weights(p, True), comp.weights(p, False), comp.hypervolume(p))


@pytest.mark.parametrize("name", CASES)
def test_bound_model_reproduces_reference_weights(name):
    case = GoldenCase(name)
    comp, bm = _bound(case)
    theta = bm.theta_of(comp.weights(case.point(0), True))
    lpe, linj, norms = log_weights(bm, theta)
    with np.errstate(all="ignore"):
        ref_pe, ref_inj = np.log(case.weights_pe), np.log(case.weights_inj)
    assert np.array_equal(np.isneginf(lpe), np.isneginf(ref_pe))
    assert np.array_equal(np.isneginf(linj), np.isneginf(ref_inj))
    ok = np.isfinite(ref_pe)
    assert np.max(np.abs(lpe[ok] - ref_pe[ok])) < 2e-12
    ok = np.isfinite(ref_inj)
    assert np.max(np.abs(linj[ok] - ref_inj[ok])) < 2e-12
    # surveyed hypervolume = Z of the designated normaliser / 1e9 * Tobs (analysis.py:267)
    assert rel_err(norms[bm.vt_norm] / 1e9 * case.tobs, case.sites[next(iter(case.flagsets))]["surveyed_hypervolume"][0]) < 1e-12


@pytest.mark.parametrize("name", CASES)
def test_bound_model_sites_via_oracle_reductions(name):
    """weights from the bound model + the oracle's reductions == golden sites, all hyper-points."""
    case = GoldenCase(name)
    comp, bm = _bound(case)
    for i in range(case.n_points):
        theta = bm.theta_of(comp.weights(case.point(i), True))
        lpe, linj, norms = log_weights(bm, theta)
        got = O.hierarchical_likelihood(lpe, linj, case.total_inj, case.nobs, case.tobs, norms[bm.vt_norm], log=True, min_neff_cut=False)
        for site in ("log_likelihood", "logBFs", "log_nEffs", "log_nEff_inj", "detection_efficiency", "rate", "variance_log_likelihood"):
            assert rel_err(got[site], case.sites["log"][site][i]) < 1e-10, (name, i, site)


def test_theta_layout_and_shared_coefficients():
    case = GoldenCase("bspline_iid")
    comp, bm = _bound(case)
    # IID spin models: two spline terms share ONE coefficient block (separable.py:77-79)
    spl = [t for t in bm.terms if t["kind"] == N.TERM_EXP_SPLINE]
    offs = [t["coef_off"] for t in spl]
    assert len(spl) == 5 and len(set(offs)) == 3
    assert bm.n_theta == 30 + 1 + 16 + 16 + 1
    # terms are in canonical (sorted-by-kind) order
    kinds = [t["kind"] for t in bm.terms]
    assert kinds == sorted(kinds)


def test_column_sharing_and_count():
    case = GoldenCase("plpeak")
    _, bm = _bound(case)
    # log m1 (shared by PL+Peak and the q power law), log q, log(1+z), kappa: SURVEY.md 8(d): C = 4 for config 2 (the kernel
    # forms m1 = exp(log m1) for the Gaussian peak itself)
    assert len(bm.pe_cols) == 4
    case = GoldenCase("bspline_full")
    _, bm = _bound(case)
    assert len(bm.pe_cols) == 9  # C = 9 for config 5


def test_structure_mismatch_is_rejected():
    case = GoldenCase("pl_test")
    comp = COMPOSITIONS["pl_test"](case.pe, case.inj)
    p = case.point(0)
    with pytest.raises(ValueError):
        bind(comp.weights(p, False), comp.weights(p, True))  # sides swapped
    with pytest.raises(TypeError):
        bind(np.ones((2, 2)), comp.weights(p, False))


def test_shard_bounds_partition():
    for n, w in ((69, 8), (200, 8), (5, 8), (100000, 3)):
        spans = [shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def test_named_theta_roundtrip():
    case = GoldenCase("bspline_full")
    comp = COMPOSITIONS["bspline_full"](case.pe, case.inj)
    bm = bind(comp.weights(case.point(1), True), comp.weights(case.point(1), False), comp.hypervolume(case.point(1)))
    comp._engine = type("E", (), {"bound": bm})()  # layout only; no GPU needed
    th = comp.theta(case.point(1))
    assert np.array_equal(th, bm.theta_of(comp.weights(case.point(1), True)))
    g = comp.named_gradient(np.arange(len(th), dtype=float))
    assert set(g) == set(comp.PARAMS) and g["m1_coefs"].shape == (30,)


def test_distribution_log_prob_faces_bind_to_the_reference_values():
    """gwinferno_amd.numpyro_distributions: what the host hands to the engine for Powerlaw (bounds as
    hyper-parameters), PowerlawRedshift and BSplineDistribution reproduces the reference's log_prob arrays
    (tests/golden/terms.npz, dist/*) -- static fractional grid indices, grid tables, normalisers."""
    import os

    from bound_eval import log_weights
    from golden_util import GOLDEN_DIR

    from gwinferno_amd import interpolation as I
    from gwinferno_amd import numpyro_distributions as D
    from gwinferno_amd.engine import bind

    z = np.load(os.path.join(GOLDEN_DIR, "terms.npz"))

    def both(make, pe_x, inj_x):
        wp, wi = make().log_prob(pe_x), make().log_prob(inj_x)  # a new distribution object per call, as analysis.py:381-399
        bm = bind(wp, wi)
        return log_weights(bm, bm.theta_of(wp))

    def check(got, ref):
        dead = ref < -1e300  # nan_to_num(-inf)
        assert np.array_equal(np.isneginf(got), dead)
        assert np.max(np.abs(got[~dead] - ref[~dead])) < 1e-11

    x = z["m1"]
    for tag, a in zip(("a", "b", "neg1", "zero"), z["powerlaw_alphas"]):
        lpe, linj, _ = both(lambda: D.Powerlaw(a, 5.0, 100.0), x.reshape(4, -1), x)
        check(linj, z[f"dist/powerlaw/{tag}"])
        check(lpe.ravel(), z[f"dist/powerlaw/{tag}"])
    zg, dv, zinj = z["dist/z_grid"], z["dist/z_dVcdz"], z["z_inj"]
    for i, lamb in enumerate(z["z_lamb"]):
        lpe, linj, norms = both(lambda: D.PowerlawRedshift(lamb, float(z["dist/powerlaw_redshift/maximum"]), zg, dv), zinj.reshape(4, -1), zinj)
        check(linj, z["dist/powerlaw_redshift/inj"][i])
        assert abs(norms[0] / z["dist/powerlaw_redshift/norm"][i] - 1) < 1e-13
    v, cs = z["dist/bspline/value"], z["dist/bspline/cs"]
    gr, grx = np.linspace(0, 1, 1000), np.linspace(0.001, 1, 1000)
    for tag, basis, g in (("bspline", I.BSpline(20, normalize=True), gr), ("logy", I.LogYBSpline(20, normalize=True), gr), ("logx", I.LogXBSpline(20, normalize=True), grx),
                          ("logxy", I.LogXLogYBSpline(20, xrange=(0.001, 1), normalize=True), grx)):
        dm = basis.bases(g)
        lpe, linj, norms = both(lambda: D.BSplineDistribution(g[0], g[-1], cs, g, dm), v.reshape(4, -1), v)
        check(linj, z[f"dist/bspline/{tag}"])
        check(lpe.ravel(), z[f"dist/bspline/{tag}"])
        assert abs(norms[0] / z[f"dist/bspline/{tag}_norm"] - 1) < 1e-13
    with pytest.raises(TypeError):
        D.BSplineDistribution(0.0, 1.0, cs, gr, np.asarray(I.BSpline(20).bases(gr)))  # a bare matrix: origin unknown
    with pytest.raises(ValueError):
        D.BSplineDistribution(0.0, 1.0, cs, grx, I.BSpline(20).bases(gr))


def test_precompile_command_validates_a_kind_sequence():
    """`python -m gwinferno_amd.precompile KIND ...` (a command line over gwi_jit_compile): the sequence checks and the
    samples-per-lane rule the engine itself applies; the compilation itself is covered by tests/test_jit_cache_cpu.py."""
    from gwinferno_amd import precompile as P

    assert P.check_kinds([2, 3, 6, 7, 7, 7, 7]) == [2, 3, 6, 7, 7, 7, 7] and P.default_samples_per_lane([2, 3, 6, 7, 7, 7, 7]) == 2
    assert P.default_samples_per_lane([3, 6, 7, 7, 7, 7, 7, 7]) == 1  # six splines: one sample per lane
    assert P.check_kinds([5, 5, 6, 107, 107], mfma=True) == [5, 5, 6, 107, 107]
    for bad in ([7, 3], [0], [15], list(range(1, 14))):
        with pytest.raises(ValueError):
            P.check_kinds(bad)


def test_ratio_term_takes_log_m1_from_the_mass_spline(monkeypatch):
    """BSplinePrimaryPowerlawRatio (config 3): the mass-ratio power law needs log m1, the m1 spline's coordinate IS log m1 (parked
    at the domain's lower edge for excluded samples): the binder hands the ratio term the spline's column with the knots as
    constants (GWI_RATIO_LOGM_FROM_SPLINE) -- 8 columns = SURVEY 8(d)'s C for configs 3/4 -- and the flat description still
    reproduces the reference's weights; GWI_FOLD_LOGM=0 keeps the separate column."""
    case = GoldenCase("bspline_iid")
    comp, bm = _bound(case)
    (ratio,) = [t for t in bm.terms if t["kind"] == N.TERM_POWERLAW_RATIO]
    m1_spline = [t for t in bm.terms if t["kind"] == N.TERM_EXP_SPLINE][0]
    assert ratio["flags"] & N.RATIO_LOGM_FROM_SPLINE and ratio["cols"][1] == m1_spline["cols"][0] and len(bm.pe_cols) == 8
    lo, hi, n_int = m1_spline["p"][0], m1_spline["p"][1], m1_spline["n_basis"] - 3
    assert ratio["p"][1:] == (lo, n_int / (hi - lo), (hi - lo) / n_int)
    theta = bm.theta_of(comp.weights(case.point(0), True))
    lpe, linj, _ = log_weights(bm, theta)
    monkeypatch.setenv("GWI_FOLD_LOGM", "0")
    _, bm0 = _bound(case)
    assert len(bm0.pe_cols) == 9 and not any(t["flags"] & N.RATIO_LOGM_FROM_SPLINE for t in bm0.terms if t["kind"] == N.TERM_POWERLAW_RATIO)
    lpe0, linj0, _ = log_weights(bm0, theta)
    assert np.array_equal(lpe, lpe0) and np.array_equal(linj, linj0)


def _array_golden():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "array_weights.npz"))


@pytest.mark.parametrize("log", [False, True])
def test_array_valued_weights_bind_to_a_unit_factor(log):
    """per_event_log_bayes_factors / detection_efficiency / hierarchical_likelihood take plain arrays in the reference
    (analysis.py:50-163).  The drop-ins wrap an array as exp(kappa) x a unit factor (likelihood._array_density): the bound
    model's per-sample log-weights are the array's logarithms, zeros (log: -inf) excluded, and the oracle's reductions on
    them give the reference's numbers for the same arrays (golden from the unmodified reference functions)."""
    from gwinferno_amd.likelihood import _array_density

    g = _array_golden()
    with np.errstate(all="ignore"):
        a, b = (g["lw_pe"], g["lw_inj"]) if log else (np.exp(g["lw_pe"]), np.exp(g["lw_inj"]))
    pw, iw = _array_density(a, log), _array_density(b, log)
    assert pw.side == "pe" and iw.side == "inj"
    bm = bind(pw, iw, None)
    assert [t["kind"] for t in bm.terms] == [N.TERM_POWERLAW] and bm.n_theta == 1
    theta = bm.theta_of(pw)
    assert np.array_equal(theta, [0.0])
    lpe, linj, _ = log_weights(bm, theta)
    assert np.array_equal(np.isneginf(lpe), np.isneginf(g["lw_pe"])) and np.array_equal(np.isneginf(linj), np.isneginf(g["lw_inj"]))
    ok = np.isfinite(g["lw_pe"])
    assert np.max(np.abs(lpe[ok] - g["lw_pe"][ok])) < 1e-13
    tag = "log" if log else "lin"
    got = O.hierarchical_likelihood(lpe, linj, float(g["total_inj"]), lpe.shape[0], 1.0, float(g["hypervolume"]) , log=True, min_neff_cut=False)
    for site in ("log_likelihood", "logBFs", "log_nEffs", "log_nEff_inj", "detection_efficiency"):
        assert rel_err(got[site], g[f"{tag}/hl_nocut/{site}"]) < 1e-10, site
    assert rel_err(got["logBFs"], g[f"{tag}/pe/logBFs"]) < 1e-10
    assert rel_err(np.log(got["detection_efficiency"]), g[f"{tag}/inj/logmu"]) < 1e-10


def test_pspline_coefficient_prior_mirror():
    """numpyro_distributions.py:302-325: log_prob = apply_difference_prior(value, inv_var, diff_order); golden from the
    reference's own class.  (Where numpyro is installed the mirror IS a numpyro Distribution; here it is the plain object.)"""
    from gwinferno_amd.numpyro_distributions import PSplineCoeficientPrior

    g = _array_golden()
    for i in range(3):
        n, tau, order = g[f"pspline/{i}/args"]
        d = PSplineCoeficientPrior(int(n), float(tau), diff_order=int(order))
        assert abs(float(d.log_prob(g[f"pspline/{i}/coefs"])) - float(g[f"pspline/{i}/log_prob"])) < 1e-13 * max(1.0, abs(float(g[f"pspline/{i}/log_prob"])))
        assert d.event_shape == (int(n),) and np.array_equal(d.sample(None, sample_shape=(2,)), np.ones(2))
        with pytest.raises(AssertionError):
            d.log_prob(np.zeros(int(n) + 1))
    assert PSplineCoeficientPrior(20, 1.0).diff_order == 2  # the reference's default


@pytest.mark.parametrize("log", [False, True])
def test_equal_weight_arrays_key_the_same_engine(log):
    """The reference calls per_event_log_bayes_factors / detection_efficiency once per likelihood evaluation with a freshly
    computed array.  The engine caches key on identities of source arrays, so `_array_density` interns equal arrays (content
    hash) and shares one zeros column per shape: a second call with an equal array -- another object -- must produce the keys
    of the first (no new engine, no eviction), and a different array must not."""
    from gwinferno_amd import likelihood as L
    from gwinferno_amd.engine import structure_key
    from gwinferno_amd.lazy import static_key

    L.clear_engine_cache()
    g = _array_golden()
    with np.errstate(all="ignore"):
        a = g["lw_pe"] if log else np.exp(g["lw_pe"])

    def one_sided_key(d):
        return (d.side, tuple(f.structure() + tuple(c.key() for c in f.columns) for f in d.factors), tuple((sgn, static_key(x)) for sgn, x in d.log_static))

    first, again, other = L._array_density(a, log), L._array_density(a.copy(), log), L._array_density(a + (0.5 if log else 1e-3), log)
    assert one_sided_key(first) == one_sided_key(again) != one_sided_key(other)
    with np.errstate(all="ignore"):
        b = g["lw_inj"] if log else np.exp(g["lw_inj"])
    assert structure_key(first, L._array_density(b, log)) == structure_key(again, L._array_density(b.copy(), log))
    L.clear_engine_cache()
