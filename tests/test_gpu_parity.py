"""GPU (-m gpu): the HIP engine, called through the C ABI, against
  (1) the golden vectors produced by the unmodified reference (tests/golden/case_*.npz):
      every hierarchical_likelihood site for every hyper-point and flag set, per-sample weights,
      and the 4th-order finite-difference gradients;
  (2) the NumPy oracle on seeded mid-size catalogs;
  (3) size-independent properties at the BASELINE sizes.
Tolerances (fp64): 1e-9 relative on values (BASELINE.json north_star); gradients vs the reference's own
finite differences 1e-9 (2e-7 for the PL+Peak mixture parameters, where the differences' truncation error dominates:
golden_util.fd_gradient_tolerance)."""
import os

import numpy as np
import pytest
from golden_util import CASES, GoldenCase, fd_gradient_tolerance, rel_err

pytestmark = pytest.mark.gpu

VALUE_RTOL = 1e-9


def _same_grad(a, b, exact):
    """Gradients of two executions of the same evaluation.  Values and scalar-parameter gradients are reduced in a fixed
    order; spline-coefficient numerators are LDS atomics from four wavefronts, whose order varies from launch to launch
    (last-bit differences) unless the engine runs in replay mode (GWI_DETERMINISTIC=1), where every bit repeats."""
    return np.array_equal(a, b) if exact else np.allclose(a, b, rtol=1e-12, atol=1e-13)


def _engine(case):
    from gwinferno_amd.compositions import COMPOSITIONS

    comp = COMPOSITIONS[case.composition](case.pe, case.inj, mmin=case.meta["mmin"], mmax=case.meta["mmax"])
    return comp, comp.engine()


def _sites_from(res, case, opt_unscaled_rate=30.0):
    s = res.summary
    out = {
        "log_likelihood": s.log_likelihood,
        "log_l": s.log_l,
        "sum_logBFs": s.sum_logBFs,
        "selection_factor": s.selection_factor,
        "log_nEff_inj": s.log_nEff_inj,
        "detection_efficiency": np.exp(s.log_det_eff),
        "variance_log_detection_efficiency": s.variance_log_detection_efficiency,
        "variance_log_likelihood": s.variance_log_likelihood,
        "logBFs": res.log_bfs,
        "log_nEffs": res.log_neffs,
        "variance_log_BFs": res.variances,
        "surveyed_hypervolume": s.surveyed_hypervolume_norm / 1e9 * case.tobs,
    }
    out["rate"] = opt_unscaled_rate / out["detection_efficiency"] / out["surveyed_hypervolume"]
    return out


@pytest.mark.parametrize("name", CASES)
def test_sites_match_reference_golden(name):
    case = GoldenCase(name)
    comp, eng = _engine(case)
    for fs, flags in case.flagsets.items():
        flags = {k: v for k, v in flags.items() if k != "log"}  # log/linear are the same arithmetic here
        for i in range(case.n_points):
            res = eng.evaluate(comp.theta(case.point(i)), case.total_inj, want_grad=not flags.get("marginalize_selection", False), **flags)
            got = _sites_from(res, case)
            for site, val in got.items():
                ref = case.sites[fs][site][i]
                # variances are differences of O(1) quantities: compare absolutely at 1e-9 of 1/n_eff scale
                if site.startswith("variance"):
                    assert np.allclose(val, ref, rtol=1e-8, atol=1e-12), (name, fs, i, site, val, ref)
                else:
                    assert rel_err(val, ref) < VALUE_RTOL, (name, fs, i, site, val, ref)
    eng.close()


@pytest.mark.parametrize("name", CASES)
def test_per_sample_weights_match_reference(name):
    case = GoldenCase(name)
    comp, eng = _engine(case)
    lpe, linj = eng.log_weights(comp.theta(case.point(0)))
    with np.errstate(all="ignore"):
        rpe, rinj = np.log(case.weights_pe), np.log(case.weights_inj)
    for got, ref in ((lpe, rpe), (linj, rinj)):
        assert np.array_equal(np.isneginf(got), np.isneginf(ref))
        ok = np.isfinite(ref)
        assert np.max(np.abs(got[ok] - ref[ok])) < 1e-10
    eng.close()


@pytest.mark.parametrize("name", CASES)
def test_gradient_matches_finite_differences_of_reference(name):
    case = GoldenCase(name)
    comp, eng = _engine(case)
    for i, fd in case.fdgrad.items():
        res = eng.evaluate(comp.theta(case.point(i)), case.total_inj, min_neff_cut=False)
        g = comp.named_gradient(res.grad, p=case.point(i))
        for pname, ref in fd.items():
            scale = max(1.0, float(np.max(np.abs(ref))))
            assert np.max(np.abs(np.asarray(g[pname]) - ref)) < fd_gradient_tolerance(name, pname) * scale, (name, i, pname, g[pname], ref)
    eng.close()


@pytest.mark.parametrize("comp_name,n_ev,n_pe,n_inj", [("plpeak", 69, 1000, 20000), ("bspline_iid", 20, 700, 9000), ("bspline_full", 12, 1500, 15001), ("plpeak_full", 10, 1000, 5000)])
def test_against_oracle_midsize(comp_name, n_ev, n_pe, n_inj):
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle import numpy_oracle as O

    pe, inj, total = make_catalog(n_ev, n_pe, n_inj, seed=99)
    comp = COMPOSITIONS[comp_name](pe, inj)
    eng = comp.engine()
    orc = O.COMPOSITIONS[comp_name](pe, inj)
    rng = np.random.default_rng(17)
    for _ in range(3):
        p = draw_params(comp_name, rng)
        res = eng.evaluate(comp.theta(p), total, min_neff_cut=False)
        ref = orc.evaluate(p, total, min_neff_cut=False)
        assert rel_err(res.log_likelihood, ref["log_likelihood"]) < VALUE_RTOL
        assert rel_err(res.log_bfs, ref["logBFs"]) < VALUE_RTOL
        assert rel_err(res.log_neffs, ref["log_nEffs"]) < VALUE_RTOL
        assert rel_err(res.summary.log_nEff_inj, ref["log_nEff_inj"]) < VALUE_RTOL
        assert rel_err(np.exp(res.summary.log_det_eff), ref["detection_efficiency"]) < VALUE_RTOL
    eng.close()


def test_empty_and_ragged_inputs():
    """Events whose samples are all excluded, a ragged last tile, and an injection set smaller than
    one workgroup."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle import numpy_oracle as O

    pe, inj, total = make_catalog(5, 257, 100, seed=5)
    pe["mass_1"][2, :] = 1.0  # below mmin: every sample of event 2 excluded
    comp = COMPOSITIONS["plpeak"](pe, inj)
    eng = comp.engine()
    p = draw_params("plpeak", np.random.default_rng(1))
    res = eng.evaluate(comp.theta(p), total, min_neff_cut=False)
    ref = O.COMPOSITIONS["plpeak"](pe, inj).evaluate(p, total, min_neff_cut=False)
    assert res.log_bfs[2] == -np.inf and ref["logBFs"][2] == -np.inf
    ok = np.isfinite(ref["logBFs"])
    assert rel_err(res.log_bfs[ok], ref["logBFs"][ok]) < VALUE_RTOL
    assert res.log_likelihood == float(ref["log_likelihood"])  # -1.797e308 sentinel (analysis.py:284-291)
    assert np.all(res.grad == 0.0)
    eng.close()


def test_run_to_run_bit_stability():
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(16, 900, 7000, seed=8)
    comp = COMPOSITIONS["bspline_test"](pe, inj)
    eng = comp.engine()
    th = comp.theta(draw_params("bspline_test", np.random.default_rng(2)))
    a = eng.evaluate(th, total, min_neff_cut=False)
    b = eng.evaluate(th, total, min_neff_cut=False)
    assert a.log_likelihood == b.log_likelihood
    assert np.array_equal(a.log_bfs, b.log_bfs)
    # scalar-parameter gradients are reduced in a fixed order; spline-coefficient numerators go through
    # LDS atomics whose order may vary: allow last-bits differences there
    assert np.allclose(a.grad, b.grad, rtol=1e-12, atol=1e-13)
    eng.close()


@pytest.mark.parametrize("iid", [True, False])
def test_parametric_masses_with_bspline_spins(iid):
    """A product written directly against the drop-in model API, as a user of the reference would: plpeak_primary_ratio_pdf
    (parametric.py:39-46) x B-spline spin magnitudes x B-spline tilts (separable.py:17-292, IID or independent) x
    PowerlawRedshiftModel -- its kind sequence (PL+Peak, PL q, PL z, four splines) has a compiled kernel of its own.  Checked
    against the C oracle, which evaluates any bound model."""
    from gwinferno_amd.engine import NativePopulationLikelihood
    from gwinferno_amd.lazy import where_finite
    from gwinferno_amd.models import (BSplineIIDSpinMagnitudes, BSplineIIDSpinTilts, BSplineIndependentSpinMagnitudes, BSplineIndependentSpinTilts, PowerlawRedshiftModel,
                                      plpeak_primary_ratio_pdf)
    from gwinferno_amd.synthetic import make_catalog
    from oracle.c_oracle import COracle

    pe, inj, total = make_catalog(12, 900, 8000, seed=5)
    rng = np.random.default_rng(3)
    if iid:
        mag = BSplineIIDSpinMagnitudes(10, pe["a_1"], pe["a_2"], inj["a_1"], inj["a_2"], normalize=True)
        tilt = BSplineIIDSpinTilts(8, pe["cos_tilt_1"], pe["cos_tilt_2"], inj["cos_tilt_1"], inj["cos_tilt_2"], normalize=True)
    else:
        mag = BSplineIndependentSpinMagnitudes(10, 9, pe["a_1"], pe["a_2"], inj["a_1"], inj["a_2"], normalize=True)
        tilt = BSplineIndependentSpinTilts(8, 7, pe["cos_tilt_1"], pe["cos_tilt_2"], inj["cos_tilt_1"], inj["cos_tilt_2"], normalize=True)
    z_model = PowerlawRedshiftModel(pe["redshift"], inj["redshift"])

    def weights(p, d, flag):
        p_m = plpeak_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], alpha=p["alpha"], beta=p["beta"], mmin=5.0, mmax=100.0, mpp=p["mpp"], sigpp=p["sigpp"], lam=p["lam"])
        p_a = mag(p["a"], pe_samples=flag) if iid else mag(p["a"], p["a2"], pe_samples=flag)
        p_t = tilt(p["t"], pe_samples=flag) if iid else tilt(p["t"], p["t2"], pe_samples=flag)
        return where_finite(p_m * p_a * p_t * z_model(d["redshift"], p["lamb"]) / d["prior"])

    def draw():
        return dict(alpha=rng.normal(-2.5, 1.0), beta=rng.normal(1.0, 1.0), mpp=rng.uniform(20.0, 50.0), sigpp=rng.uniform(1.0, 10.0), lam=rng.uniform(0.0, 0.2), lamb=rng.normal(2.7, 1.0),
                    a=rng.normal(size=10), a2=rng.normal(size=9), t=rng.normal(size=8), t2=rng.normal(size=7))

    p0 = draw()
    wpe = weights(p0, pe, True)
    eng = NativePopulationLikelihood(wpe, weights(p0, inj, False), z_model.normalization(p0["lamb"]))
    orc = COracle(eng.bound)
    for _ in range(4):
        p = draw()
        th = eng.bound.theta_of(weights(p, pe, True))
        got = eng.evaluate(th, total, min_neff_cut=False)
        ref = orc.evaluate(th, total, min_neff_cut=False)
        assert rel_err(got.log_likelihood, ref["log_likelihood"]) < VALUE_RTOL
        assert rel_err(got.log_bfs, ref["logBFs"]) < VALUE_RTOL
        scale = max(1.0, float(np.max(np.abs(ref["grad"]))))
        assert float(np.max(np.abs(got.grad - ref["grad"]))) / scale < 1e-8
    eng.close()


@pytest.mark.parametrize("comp_name", ["bspline_test", "bspline_full", "bspline_chieff", "chm_bspline", "bspline_misc"])  # bspline_misc: a PL+Peak term that absorbs the closing exponential, in a spline chain
def test_replay_mode_is_bit_reproducible(comp_name, monkeypatch):
    """GWI_DETERMINISTIC=1: the shared gradient rows are filled in one fixed order -- one replica per
    lane, the wavefronts of a workgroup take turns -- so every bit of the value, the sites AND the spline-coefficient
    gradient repeats from run to run and from engine to engine (single and batched launches); the result agrees with
    the regular (atomic, order-free) mode to rounding."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(16, 2100, 9000, seed=8)
    rng = np.random.default_rng(2)
    fast = COMPOSITIONS[comp_name](pe, inj)
    ths = np.stack([fast.theta(draw_params(comp_name, rng)) for _ in range(3)])
    ref = [fast.engine().evaluate(th, total, min_neff_cut=False) for th in ths]
    monkeypatch.setenv("GWI_DETERMINISTIC", "1")
    runs = []
    for _ in range(2):  # two engines built independently
        eng = COMPOSITIONS[comp_name](pe, inj).engine()
        one = [eng.evaluate(th, total, min_neff_cut=False) for th in ths]
        again = [eng.evaluate(th, total, min_neff_cut=False) for th in ths[::-1]][::-1]
        batch = eng.evaluate_batch(ths, total, min_neff_cut=False)
        for a, b, c in zip(one, again, batch):
            assert a.log_likelihood == b.log_likelihood == c.log_likelihood
            assert np.array_equal(a.grad, b.grad) and np.array_equal(a.grad, c.grad)
            assert np.array_equal(a.log_bfs, b.log_bfs) and np.array_equal(a.log_neffs, b.log_neffs)
        runs.append(one)
        eng.close()
    for a, b, r in zip(runs[0], runs[1], ref):
        assert a.log_likelihood == b.log_likelihood and np.array_equal(a.grad, b.grad)
        assert rel_err(a.log_likelihood, r.log_likelihood) < 1e-13
        assert np.allclose(a.grad, r.grad, rtol=1e-11, atol=1e-12)
    fast.engine().close()


def _spline_shift(comp, theta, amount):
    """theta with `amount` added to every coefficient of the first exp-spline term: B-splines are a partition of unity, so
    every log-weight inside that term's domain moves by exactly `amount` (the grid normaliser moves with it)."""
    from gwinferno_amd import _native as N

    t = next(t for t in comp.engine().bound.terms if t["kind"] == N.TERM_EXP_SPLINE)
    out = np.array(theta, dtype=float)
    out[t["coef_off"] : t["coef_off"] + t["n_basis"]] += amount
    return out


@pytest.mark.parametrize("comp_name", ["bspline_test", "bspline_iid", "bspline_misc"])
def test_reference_exponent_outrun_triggers_one_repeat(comp_name, monkeypatch):
    """Spline models weigh a tile against the tile's exact maximum at the PREVIOUS evaluation of the handle (0 before the
    first).  The result does not depend on that reference to the bit; what can go wrong is the range, and then the scan
    asks for ONE repeat, which finds the exact maxima in place.  Forced here three ways: a sampling prior 1e200 times
    larger everywhere (first evaluation: every log-weight 460 below the initial reference 0), a jump of +-400 in all
    coefficients of one spline between two evaluations, and the same on the batched and begin/end paths.  What used to
    need the repeat -- weights spanning 184 e-folds INSIDE a tile -- no longer does."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle.c_oracle import COracle

    pe, inj, total = make_catalog(9, 4000, 30000, seed=12)
    comp0 = COMPOSITIONS[comp_name](pe, inj)
    plain = comp0.engine()
    ths = [comp0.theta(draw_params(comp_name, np.random.default_rng(4 + k))) for k in range(3)]
    first = plain.evaluate(ths[0], total, min_neff_cut=False)
    assert plain.two_pass_repeats() == 0
    plain.evaluate(ths[1], total, min_neff_cut=False)
    again = plain.evaluate(ths[0], total, min_neff_cut=False)  # other references this time: same bits (atomics aside)
    assert plain.two_pass_repeats() == 0
    assert again.log_likelihood == first.log_likelihood and np.array_equal(again.log_bfs, first.log_bfs) and np.array_equal(again.log_neffs, first.log_neffs)
    assert np.allclose(again.grad, first.grad, rtol=1e-12, atol=1e-13)
    plain.close()

    def check(got_ll, got_grad, ref, got_bfs=None):
        assert rel_err(got_ll, ref["log_likelihood"]) < VALUE_RTOL
        if got_bfs is not None:
            assert rel_err(got_bfs, ref["logBFs"]) < VALUE_RTOL
        scale = max(1.0, float(np.max(np.abs(ref["grad"]))))
        assert float(np.max(np.abs(np.asarray(got_grad) - ref["grad"]))) / scale < 1e-8

    pe2 = {k: v.copy() for k, v in pe.items()}
    inj2 = {k: v.copy() for k, v in inj.items()}
    pe2["prior"] *= 1e200
    inj2["prior"] *= 1e200
    # ... and inside every tile of 2048 samples the first 512 another 1e80 on top
    pe2["prior"][:, (np.arange(pe2["prior"].shape[1]) % 2048) < 512] *= 1e80
    inj2["prior"][(np.arange(inj2["prior"].shape[0]) % 2048) < 512] *= 1e80
    monkeypatch.setenv("GWI_SAMPLES_PER_BLOCK", "2048")
    comp = COMPOSITIONS[comp_name](pe2, inj2)
    eng = comp.engine()
    orc = COracle(eng.bound)
    for k in range(3):
        got = eng.evaluate(ths[k], total, min_neff_cut=False)
        assert eng.two_pass_repeats() == 1  # the first evaluation only
        ref = orc.evaluate(ths[k], total, min_neff_cut=False)
        check(got.log_likelihood, got.grad, ref, got.log_bfs)
        assert rel_err(got.log_neffs, ref["log_nEffs"]) < 1e-8
    # a jump in theta that moves every log-weight by +400, then back: one repeat each
    for n, amount in enumerate((400.0, 0.0)):
        th = _spline_shift(comp, ths[0], amount)
        got = eng.evaluate(th, total, min_neff_cut=False)
        assert eng.two_pass_repeats() == 2 + n
        check(got.log_likelihood, got.grad, orc.evaluate(th, total, min_neff_cut=False), None)
    # marginalised selection: the squared-weight pass and the regular pass each report their own exponent
    n0 = eng.two_pass_repeats()
    th = _spline_shift(comp, ths[1], -350.0)
    got = eng.evaluate(th, total, min_neff_cut=False, marginalize_selection=True)
    assert eng.two_pass_repeats() == n0 + 1
    ref = orc.evaluate(th, total, min_neff_cut=False, marginalize_selection=True)
    assert rel_err(got.log_likelihood, ref["log_likelihood"]) < VALUE_RTOL
    if "grad" in ref and ref["grad"] is not None:
        scale = max(1.0, float(np.max(np.abs(ref["grad"]))))
        assert float(np.max(np.abs(got.grad - ref["grad"]))) / scale < 1e-8
    # the batched launch (references per point) and the begin / end pair take the same detour
    tb = np.stack([ths[0], _spline_shift(comp, ths[1], 380.0), ths[2]])
    n0 = eng.two_pass_repeats()
    batch = eng.evaluate_batch(tb, total, min_neff_cut=False)
    assert eng.two_pass_repeats() == n0 + 1  # first batched launch of the handle: its rows hold no references yet
    batch = eng.evaluate_batch(tb, total, min_neff_cut=False)
    assert eng.two_pass_repeats() == n0 + 1
    begin, end = eng.configure_async(total, min_neff_cut=False)
    for k in range(3):
        ref = orc.evaluate(tb[k], total, min_neff_cut=False)
        check(batch[k].log_likelihood, batch[k].grad, ref)
        begin(tb[k])
        ll, g = end()
        check(ll, g, ref)
    assert eng.two_pass_repeats() == n0 + 4  # the batch's first launch, then -350 -> 0 -> +380 -> 0 on the single-evaluation row
    eng.close()


def test_wide_prior_random_walk_needs_no_repeats():
    """200 evaluations of the config-3 model along a random walk that starts at a N(0, 10) draw of
    every spline coefficient and moves every coefficient by N(0, 0.7) per evaluation -- far more than a leapfrog step --
    never leave the range around the previous evaluation's tile maxima: at most the first evaluation is repeated.  Spot
    checks against the C oracle along the way."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle.c_oracle import COracle

    pe, inj, total = make_catalog(12, 3000, 20000, seed=5)
    comp = COMPOSITIONS["bspline_iid"](pe, inj)
    eng = comp.engine()
    orc = COracle(eng.bound)
    rng = np.random.default_rng(11)
    base = comp.theta(draw_params("bspline_iid", rng))
    from gwinferno_amd import _native as N

    coef = np.zeros(eng.n_theta, dtype=bool)
    for t in eng.bound.terms:
        if t["kind"] == N.TERM_EXP_SPLINE:
            coef[t["coef_off"] : t["coef_off"] + t["n_basis"]] = True
    th = base.copy()
    th[coef] = 10.0 * rng.standard_normal(int(coef.sum()))
    n = 200
    for i in range(n):
        got = eng.evaluate(th, total, min_neff_cut=False)
        if i % 50 == 0:
            ref = orc.evaluate(th, total, min_neff_cut=False)
            assert rel_err(got.log_likelihood, ref["log_likelihood"]) < VALUE_RTOL
            scale = max(1.0, float(np.max(np.abs(ref["grad"]))))
            assert float(np.max(np.abs(got.grad - ref["grad"]))) / scale < 1e-8
        th[coef] += 0.7 * rng.standard_normal(int(coef.sum()))
    assert eng.two_pass_repeats() <= 1, eng.two_pass_repeats()
    eng.close()


@pytest.mark.parametrize("comp_name", ["bspline_test", "chm_powerlaw", "chm_bspline"])
def test_partial_records_combine_like_single_device(comp_name):
    """Two shards evaluated on the same GPU and combined == the unsharded evaluation (the
    multi-GPU path minus the RCCL exchange)."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.engine import NativePopulationLikelihood
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(9, 500, 3001, seed=21)
    comp = COMPOSITIONS[comp_name](pe, inj)
    p = draw_params(comp_name, np.random.default_rng(4))
    if comp_name == "chm_powerlaw":
        p["mmin"], p["mmax"] = 4.0, 110.0  # truncation bounds that leave every event some samples
    full = comp.engine()
    th = comp.theta(p)
    ref = full.evaluate(th, total, min_neff_cut=False)
    recs = []
    for r in range(2):
        e = NativePopulationLikelihood(comp.weights(p, True), comp.weights(p, False), comp.hypervolume(p), rank=r, world=2)
        recs.append(e.eval_partial(th)[0])
        last = e
    out = last.combine(np.stack(recs), total, nobs=9, min_neff_cut=False)
    assert rel_err(out.log_likelihood, ref.log_likelihood) < 1e-12
    assert np.allclose(out.grad, ref.grad, rtol=1e-10, atol=1e-11)
    full.close()


@pytest.mark.parametrize("replay", [False, True])
def test_in_engine_rccl_exchange_world1(replay, monkeypatch):
    """gwi_comm_init + gwi_eval_sharded (scan -> ncclAllGather -> publish -> assemble on one stream)
    with a one-rank communicator reproduces gwi_eval bit for bit."""
    if replay:
        monkeypatch.setenv("GWI_DETERMINISTIC", "1")
    import ctypes as C

    from gwinferno_amd import _native as N
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.distributed import rccl_library_path
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(9, 500, 3001, seed=21)
    comp = COMPOSITIONS["bspline_test"](pe, inj)
    eng = comp.engine()
    path = rccl_library_path()
    buf = C.create_string_buffer(128)
    assert N.load_library().gwi_comm_unique_id(path.encode() if path else None, buf) == 0
    eng.comm_init(buf.raw, 0, 1, rccl_path=path)
    rng = np.random.default_rng(4)
    for _ in range(3):
        th = comp.theta(draw_params("bspline_test", rng))
        a = eng.evaluate(th, total, min_neff_cut=False)
        b = eng.evaluate_sharded(th, total, min_neff_cut=False)
        # per-event values are the same arithmetic; the sums over events run on the host in one path and in
        # final_kernel in the other (different association): last-bit differences only
        assert np.array_equal(a.log_bfs, b.log_bfs)
        assert rel_err(a.log_likelihood, b.log_likelihood) < 1e-14
        assert np.allclose(a.grad, b.grad, rtol=1e-12, atol=1e-13)
        assert rel_err(a.summary.log_nEff_inj, b.summary.log_nEff_inj) < 1e-13
        # marginalised selection: the sharded path runs a second exchange for the squared-weight records
        am = eng.evaluate(th, total, min_neff_cut=False, marginalize_selection=True)
        bm_ = eng.evaluate_sharded(th, total, min_neff_cut=False, marginalize_selection=True)
        assert rel_err(am.log_likelihood, bm_.log_likelihood) < 1e-14
        assert np.allclose(am.grad, bm_.grad, rtol=1e-11, atol=1e-12)
        assert np.max(np.abs(am.grad - a.grad)) > 0
    # the in-library loop switches to the sharded entry once a communicator is attached (bench.py at N > 1)
    ths = np.stack([comp.theta(draw_params("bspline_test", rng)) for _ in range(4)])
    ll, grads = eng.evaluate_sequence(ths, total, min_neff_cut=False)
    for i, th in enumerate(ths):
        b = eng.evaluate_sharded(th, total, min_neff_cut=False)
        assert b.log_likelihood == ll[i] and _same_grad(b.grad, grads[i], replay)
    # ... and so does the library's sampler: a short chain on the sharded handle moves and stays finite
    from gwinferno_amd.sampling import GaussianSmoothingPrior, nuts_engine

    prior = GaussianSmoothingPrior(eng.n_theta).normal(slice(0, eng.n_theta), 3.0)
    (out,) = nuts_engine([eng], total, prior, None, [ths[0]], n_warmup=15, n_samples=10, max_tree_depth=4, seed=3, min_neff_cut=False)
    assert np.all(np.isfinite(out["samples"])) and np.std(out["samples"], axis=0).min() > 0 and out["n_evals"] > 25
    eng.close()


@pytest.mark.parametrize("comp_name", ["plpeak", "bspline_test", "bspline_iid", "chm_powerlaw", "chm_bspline"])
def test_batched_evaluation_matches_single(comp_name):
    """gwi_eval_batch (K points per launch, blockIdx.y = point) == K separate gwi_eval calls."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(11, 700, 5003, seed=41)
    comp = COMPOSITIONS[comp_name](pe, inj)
    eng = comp.engine()
    rng = np.random.default_rng(6)
    thetas = np.stack([comp.theta(draw_params(comp_name, rng)) for _ in range(7)])
    batch = eng.evaluate_batch(thetas, total, min_neff_cut=False)
    for k in range(len(thetas)):
        one = eng.evaluate(thetas[k], total, min_neff_cut=False)
        b = batch[k]
        assert rel_err(b.log_likelihood, one.log_likelihood) < 1e-12
        assert np.allclose(b.log_bfs, one.log_bfs, rtol=1e-12, atol=1e-12)
        assert np.allclose(b.log_neffs, one.log_neffs, rtol=1e-10)
        assert np.allclose(b.grad, one.grad, rtol=1e-10, atol=1e-11)
        assert np.allclose(b.norms, one.norms, rtol=1e-13)
        assert rel_err(b.summary.log_nEff_inj, one.summary.log_nEff_inj) < 1e-10
    # the lean entry for vectorised chains (preallocated buffers) and both theta-upload paths (K < 10: copy, K >= 10: stage kernel)
    for K in (3, 12):
        tk = np.stack([thetas[k % len(thetas)] for k in range(K)])
        values, grads = eng.configure_batch(K, total, min_neff_cut=False)(tk)
        for k in range(K):
            one = eng.evaluate(tk[k], total, min_neff_cut=False)
            assert rel_err(values[k], one.log_likelihood) < 1e-12
            assert np.allclose(grads[k], one.grad, rtol=1e-10, atol=1e-11)
    eng.close()


@pytest.mark.parametrize("comp_name", ["plpeak", "bspline_test", "chm_bspline"])
def test_batches_in_two_halves_from_one_thread(comp_name, monkeypatch):
    """gwi_eval_batch_begin / gwi_eval_batch_end: two engines on the same catalog, driven alternately from ONE thread (a set of
    points in flight on each), return bit for bit what the blocking gwi_eval_batch returns for the same points -- and the state
    machine refuses a second begin, an end without a begin, and gwi_eval_end on a pending batch."""
    from gwinferno_amd._native import NativeEngineError
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog

    monkeypatch.setenv("GWI_PBATCH_PTS", "4")  # (parametric chains: the several-points-per-workgroup kernel also at this size)
    pe, inj, total = make_catalog(11, 700, 5003, seed=41)
    comps = [COMPOSITIONS[comp_name](pe, inj) for _ in range(2)]
    engs = [c.engine() for c in comps]
    assert engs[0].handle != engs[1].handle
    rng = np.random.default_rng(8)
    K = 12
    sets = [np.stack([comps[0].theta(draw_params(comp_name, rng)) for _ in range(K)]) for _ in range(4)]
    blocking = [tuple(np.copy(a) for a in engs[0].configure_batch(K, total, min_neff_cut=False)(t)) for t in sets]
    halves = [e.configure_batch_async(K, total, min_neff_cut=False) for e in engs]
    got = [None] * len(sets)
    halves[0][0](sets[0])
    for i in range(1, len(sets)):  # set i goes out before set i - 1 is collected
        halves[i % 2][0](sets[i])
        v, g = halves[(i - 1) % 2][1]()
        got[i - 1] = (v.copy(), g.copy())
    v, g = halves[(len(sets) - 1) % 2][1]()
    got[-1] = (v.copy(), g.copy())
    for (bv, bg), (av, ag) in zip(blocking, got):
        assert np.array_equal(bv, av)
        assert np.array_equal(bg, ag) or np.allclose(bg, ag, rtol=1e-13, atol=1e-13)  # (LDS atomics: the 4-tap kernel's gradient may differ in its last bits from run to run)
    # the state machine
    begin, end = halves[0]
    with pytest.raises(NativeEngineError, match="without gwi_eval_batch_begin"):
        end()
    begin(sets[0])
    with pytest.raises(NativeEngineError):
        begin(sets[1])
    with pytest.raises(NativeEngineError):
        engs[0].evaluate(sets[0][0], total, min_neff_cut=False)
    st = engs[0].lib.gwi_eval_end(engs[0].handle, None, None, None, None, None, None)
    assert st == -1  # GWI_ERR_INVALID
    v, g = end()
    assert np.array_equal(v, blocking[0][0])
    one = engs[0].evaluate(sets[0][3], total, min_neff_cut=False)  # and the handle is free again
    assert rel_err(one.log_likelihood, v[3]) < 1e-12
    for e in engs:
        e.close()


@pytest.mark.parametrize("path", ["mfma", "rows"])
@pytest.mark.parametrize("comp_name", ["bspline_test", "bspline_iid", "bspline_full", "bspline_defaults", "bspline_chieff"])
def test_batched_launch_on_the_matrix_cores(comp_name, path, monkeypatch):
    """The MFMA path of gwi_eval_batch (gwinferno_amd/csrc/gwi_mfma.h: spline-coefficient gradient as a
    v_mfma_f64_16x16x4 GEMM over 16 hyper-parameter points per wavefront; reverse mode of interpolation.py:304) against
    the single-point kernel, the 4-tap batched kernel and the C oracle -- full and ragged groups of 16."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle.c_oracle import COracle

    env_name = {"mfma": "GWI_BATCH_MFMA", "rows": "GWI_BATCH_ROWS"}[path]
    monkeypatch.setenv("GWI_MAX_BATCH", "40")
    monkeypatch.setenv(env_name, "1")
    pe, inj, total = make_catalog(11, 700, 5003, seed=41)
    rng = np.random.default_rng(6)
    comp = COMPOSITIONS[comp_name](pe, inj)
    eng = comp.engine()
    assert eng.batch_path(16) == path and eng.batch_path(4) == "taps"
    # (the default, with neither variable set: "mfma", or "rows" for models with more than 8 gradient tiles -- bspline_defaults)
    thetas = np.stack([comp.theta(draw_params(comp_name, rng)) for _ in range(35)])
    orc = COracle(eng.bound)
    refs = [orc.evaluate(t, total, min_neff_cut=False) for t in thetas[:6]]
    singles = [eng.evaluate(t, total, min_neff_cut=False) for t in thetas]
    for K in (9, 16, 17, 35):
        batch = eng.evaluate_batch(thetas[:K], total, min_neff_cut=False)
        again = eng.evaluate_batch(thetas[:K], total, min_neff_cut=False)
        for k in range(K):
            b, one = batch[k], singles[k]
            assert rel_err(b.log_likelihood, one.log_likelihood) < 1e-12
            assert np.allclose(b.log_bfs, one.log_bfs, rtol=1e-12, atol=1e-12)
            assert np.allclose(b.log_neffs, one.log_neffs, rtol=1e-10)
            assert np.allclose(b.grad, one.grad, rtol=1e-10, atol=1e-11)
            assert np.allclose(b.norms, one.norms, rtol=1e-13)
            # registers and a fixed order all the way: the matrix-core path repeats bit for bit (the LDS rows take atomics
            # from four wavefronts: last bits of the gradient may differ)
            assert b.log_likelihood == again[k].log_likelihood
            assert np.array_equal(b.grad, again[k].grad) if path == "mfma" else np.allclose(b.grad, again[k].grad, rtol=1e-12, atol=1e-13)
            if k < len(refs):
                r = refs[k]
                assert rel_err(b.log_likelihood, r["log_likelihood"]) < VALUE_RTOL
                assert rel_err(b.log_bfs, r["logBFs"]) < VALUE_RTOL
                scale = max(1.0, float(np.max(np.abs(r["grad"]))))
                assert float(np.max(np.abs(b.grad - r["grad"]))) / scale < 1e-8
    eng.close()
    # the same batch through the 4-tap kernel
    monkeypatch.delenv(env_name)
    monkeypatch.setenv("GWI_BATCH_MFMA", "0")
    eng2 = COMPOSITIONS[comp_name](pe, inj).engine()
    assert eng2.batch_path(16) == "taps"
    taps = eng2.evaluate_batch(thetas[:16], total, min_neff_cut=False)
    for k in range(16):
        assert rel_err(taps[k].log_likelihood, singles[k].log_likelihood) < 1e-12
        assert np.allclose(taps[k].grad, singles[k].grad, rtol=1e-10, atol=1e-11)
    eng2.close()


@pytest.mark.parametrize("comp_name", ["plpeak", "bspline_test", "bspline_iid", "bspline_full"])
def test_gradient_with_marginalised_selection(comp_name):
    """marginalize_selection=True (analysis.py:270-271) adds -(3+N_obs)/(2 n_eff_inj) to log mu; its
    gradient needs sum_j w_j^2 dl_j/dtheta, which the engine takes from a second, squared-weight pass.
    Held at 1e-8 against the C oracle's analytic gradient, which tests/test_c_oracle.py pins to finite differences of the
    unmodified reference under this flag (tests/golden/margsel_grad.npz); batched == single."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle.c_oracle import COracle

    # few injections: n_eff_inj is small, so the marginalisation term carries real weight in the gradient
    pe, inj, total = make_catalog(9, 300, 700, seed=12)
    comp = COMPOSITIONS[comp_name](pe, inj)
    eng = comp.engine()
    orc = COracle(eng.bound)
    rng = np.random.default_rng(8)
    flags = dict(marginalize_selection=True, min_neff_cut=False)
    for _ in range(3):
        th = comp.theta(draw_params(comp_name, rng))
        res = eng.evaluate(th, total, **flags)
        plain = eng.evaluate(th, total, min_neff_cut=False)
        ref = orc.evaluate(th, total, **flags)
        assert rel_err(res.log_likelihood, ref["log_likelihood"]) < VALUE_RTOL
        assert np.max(np.abs(res.grad - plain.grad)) > 1e-6  # the extra term is not negligible here
        scale = max(1.0, float(np.max(np.abs(ref["grad"]))))
        assert float(np.max(np.abs(res.grad - ref["grad"]))) / scale < 1e-8
    thetas = np.stack([comp.theta(draw_params(comp_name, rng)) for _ in range(3)])
    batch = eng.evaluate_batch(thetas, total, **flags)
    for k in range(3):
        one = eng.evaluate(thetas[k], total, **flags)
        assert np.allclose(batch[k].grad, one.grad, rtol=1e-10, atol=1e-11)
    eng.close()


@pytest.mark.parametrize("n_ev,n_pe,n_inj,env", [
    (700, 40, 3000, {}),                                   # many short events: one partly filled wave per event
    (3, 20000, 300000, {"GWI_SAMPLES_PER_BLOCK": "512"}),  # > 64 injection groups -> widened groups; 40 tiles per event
    (6, 33, 40, {}),                                       # everything shorter than one wave
    (64, 1, 65, {}),                                       # one posterior sample per event
])
def test_launch_geometry_extremes(n_ev, n_pe, n_inj, env, monkeypatch):
    """Shapes that stress the tile / group bookkeeping (tiles per event, injection groups > 64 ->
    regrouped, device-final vs host-final mode) against the NumPy oracle."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle import numpy_oracle as O

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pe, inj, total = make_catalog(n_ev, n_pe, n_inj, seed=23)
    comp = COMPOSITIONS["bspline_test"](pe, inj)
    eng = comp.engine()
    orc = O.COMPOSITIONS["bspline_test"](pe, inj)
    p = draw_params("bspline_test", np.random.default_rng(4))
    res = eng.evaluate(comp.theta(p), total, min_neff_cut=False)
    ref = orc.evaluate(p, total, min_neff_cut=False)
    ok = np.isfinite(ref["logBFs"])
    assert np.array_equal(np.isfinite(res.log_bfs), ok)
    assert rel_err(res.log_bfs[ok], ref["logBFs"][ok]) < VALUE_RTOL
    # log n_eff = 2 logsumexp(l) - logsumexp(2 l) is exactly 0 for a single sample in the reference's form; here it is the
    # difference of two logs of a normalised sum and its square (1e-16): absolute tolerance next to the relative one
    assert np.allclose(res.log_neffs[ok], ref["log_nEffs"][ok], rtol=VALUE_RTOL, atol=1e-13)
    assert rel_err(np.exp(res.summary.log_det_eff), ref["detection_efficiency"]) < VALUE_RTOL
    if n_inj > 1:
        assert rel_err(res.summary.log_nEff_inj, ref["log_nEff_inj"]) < 1e-8
    if np.isfinite(float(ref["log_likelihood"])) and abs(float(ref["log_likelihood"])) < 1e300:
        assert rel_err(res.log_likelihood, ref["log_likelihood"]) < VALUE_RTOL
    eng.close()


def test_few_events_with_very_many_posterior_samples_need_no_knob():
    """3 events x 1 M posterior samples (+ 50 k injections) with NO environment knob: the default tile size would give an
    event 490 tile records, more than the 64 one wave combines -- gwi_create grows the tiles instead of refusing
    (VERDICT r1 weak 7).  Checked against the C oracle: value, sites, gradient."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle.c_oracle import COracle

    for k in ("GWI_SAMPLES_PER_BLOCK", "GWI_PE_CHUNK", "GWI_INJ_CHUNK"):
        assert k not in os.environ
    pe, inj, total = make_catalog(3, 1_000_000, 50_000, seed=29)
    for name in ("plpeak", "bspline_test"):
        comp = COMPOSITIONS[name](pe, inj)
        eng = comp.engine()
        th = comp.theta(draw_params(name, np.random.default_rng(8)))
        got = eng.evaluate(th, total, min_neff_cut=False)
        ref = COracle(eng.bound).evaluate(th, total, min_neff_cut=False)
        assert rel_err(got.log_likelihood, ref["log_likelihood"]) < VALUE_RTOL
        assert rel_err(got.log_bfs, ref["logBFs"]) < VALUE_RTOL
        assert rel_err(got.log_neffs, ref["log_nEffs"]) < 1e-8
        scale = max(1.0, float(np.max(np.abs(ref["grad"]))))
        assert float(np.max(np.abs(got.grad - ref["grad"]))) / scale < 1e-8
        eng.close()


def test_a_tile_size_knob_below_the_group_limit_is_raised(monkeypatch):
    """GWI_SAMPLES_PER_BLOCK=512 on 2 events x 35 840 samples asks for 70 tiles per event: the engine raises the tile
    size to what one wave can combine (64 records) and evaluates correctly."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle import numpy_oracle as O

    monkeypatch.setenv("GWI_SAMPLES_PER_BLOCK", "512")
    pe, inj, total = make_catalog(2, 512 * 70, 100, seed=1)
    comp = COMPOSITIONS["pl_test"](pe, inj)
    p = draw_params("pl_test", np.random.default_rng(2))
    res = comp.engine().evaluate(comp.theta(p), total, min_neff_cut=False)
    ref = O.COMPOSITIONS["pl_test"](pe, inj).evaluate(p, total, min_neff_cut=False)
    assert rel_err(res.log_bfs, ref["logBFs"]) < VALUE_RTOL
    assert rel_err(res.log_likelihood, ref["log_likelihood"]) < VALUE_RTOL


def test_degenerate_samples_do_not_poison_the_gradient():
    """Samples whose weight the reference turns into 0 through its NaN/Inf guard (tests/inference_test.py:172):
    m1 exactly at mmin (empty q interval), q = 0, a = 1, NaN data.  They must count as zero weight AND leave the
    gradient finite (the kernel carries per-sample gradient state; 0 x NaN would be NaN)."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle import numpy_oracle as O

    pe, inj, total = make_catalog(6, 300, 2000, seed=31)
    pe["mass_1"][0, 0] = 5.0
    pe["mass_ratio"][1, 1] = 0.0
    pe["a_1"][2, 2] = 1.0
    pe["cos_tilt_1"][3, 3] = np.nan  # (a NaN redshift would poison zmin/zmax of the reference's redshift model globally)
    pe["cos_tilt_2"][4, 4] = np.nan
    inj["mass_1"][5] = 5.0
    inj["a_2"][6] = 1.0
    comp = COMPOSITIONS["plpeak_full"](pe, inj)
    eng = comp.engine()
    orc = O.COMPOSITIONS["plpeak_full"](pe, inj)
    rng = np.random.default_rng(2)
    for _ in range(3):
        p = draw_params("plpeak_full", rng)
        th = comp.theta(p)
        res = eng.evaluate(th, total, min_neff_cut=False)
        ref = orc.evaluate(p, total, min_neff_cut=False)
        assert np.all(np.isfinite(res.grad)), res.grad
        assert rel_err(res.log_likelihood, ref["log_likelihood"]) < VALUE_RTOL
        assert rel_err(res.log_bfs, ref["logBFs"]) < VALUE_RTOL
        # the analytic gradient is the gradient of that value
        k = int(rng.integers(len(th)))
        h = 1e-5 * max(1.0, abs(th[k]))
        e = np.zeros_like(th)
        e[k] = h
        fd = (eng.evaluate(th + e, total, min_neff_cut=False, want_grad=False).log_likelihood - eng.evaluate(th - e, total, min_neff_cut=False, want_grad=False).log_likelihood) / (2 * h)
        assert abs(res.grad[k] - fd) < 1e-5 * max(1.0, abs(fd))
    eng.close()


def test_begin_end_and_interleaved_engines():
    """gwi_eval_begin / gwi_eval_end: the two halves give the bits of gwi_eval; two engines with evaluations in
    flight together do not disturb each other; a second begin on a busy handle is refused."""
    from gwinferno_amd._native import NativeEngineError
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(9, 500, 3000, seed=19)
    comps = [COMPOSITIONS["bspline_test"](pe, inj) for _ in range(2)]
    engs = [c.engine() for c in comps]
    rng = np.random.default_rng(3)
    ths = [comps[0].theta(draw_params("bspline_test", rng)) for _ in range(6)]
    refs = [engs[0].evaluate(t, total, min_neff_cut=False) for t in ths]
    (b0, e0), (b1, e1) = (e.configure_async(total, min_neff_cut=False) for e in engs)
    for i in range(0, 6, 2):
        b0(ths[i])
        b1(ths[i + 1])
        with pytest.raises(NativeEngineError, match="has not been collected"):
            b0(ths[i])
        v1, g1 = e1()
        assert v1 == refs[i + 1].log_likelihood and np.allclose(g1, refs[i + 1].grad, rtol=1e-12, atol=1e-13)
        v0, g0 = e0()
        assert v0 == refs[i].log_likelihood and np.allclose(g0, refs[i].grad, rtol=1e-12, atol=1e-13)
    with pytest.raises(NativeEngineError, match="without gwi_eval_begin"):
        e0()
    # while an evaluation begun with gwi_eval_begin is in flight, EVERY other evaluating entry point of that
    # handle is refused (it would overwrite the kernel arguments and the sequence stamp under it) -- and the pending
    # evaluation still collects the right result afterwards
    b0(ths[0])
    for call in (lambda: engs[0].evaluate_batch(np.stack(ths[:4]), total, min_neff_cut=False), lambda: engs[0].log_weights(ths[1]), lambda: engs[0].eval_partial(ths[1]),
                 lambda: engs[0].evaluate(ths[1], total, min_neff_cut=False)):
        with pytest.raises(NativeEngineError, match="has not been collected"):
            call()
    v0, g0 = e0()
    assert v0 == refs[0].log_likelihood and np.allclose(g0, refs[0].grad, rtol=1e-12, atol=1e-13)
    for e in engs:
        e.close()


def test_engines_in_concurrent_host_threads():
    """One engine per host thread, evaluations running concurrently (handles are independent: own stream, buffers
    and catalog copy; the library releases the GIL): every thread gets exactly the single-threaded results."""
    import threading

    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(9, 500, 3000, seed=19)
    n_threads = 4
    comps = [COMPOSITIONS["bspline_test"](pe, inj) for _ in range(n_threads)]
    engs = [c.engine() for c in comps]
    rng = np.random.default_rng(8)
    ths = [comps[0].theta(draw_params("bspline_test", rng)) for _ in range(8)]
    refs = [engs[0].evaluate(t, total, min_neff_cut=False) for t in ths]
    errors = []

    def work(c):
        vg = engs[c].configure(total, min_neff_cut=False)
        for i in range(300):
            k = (i + c) % len(ths)
            v, g = vg(ths[k])
            if v != refs[k].log_likelihood or not np.allclose(g, refs[k].grad, rtol=1e-12, atol=1e-13):
                errors.append((c, i, v, refs[k].log_likelihood))
                return

    workers = [threading.Thread(target=work, args=(c,)) for c in range(n_threads)]
    for w in workers:
        w.start()
    for w in workers:
        w.join()
    assert not errors, errors[:3]
    for e in engs:
        e.close()


@pytest.mark.parametrize("replay", [False, True])
def test_eval_sequence_equals_one_by_one(replay, monkeypatch):
    """gwi_eval_sequence (the sampler's loop inside the library) returns, point by point, exactly what gwi_eval
    returns; kernel timings come back for every `timing_every`-th point only."""
    if replay:
        monkeypatch.setenv("GWI_DETERMINISTIC", "1")
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(9, 300, 3000, seed=21)
    comp = COMPOSITIONS["plpeak"](pe, inj)
    eng = comp.engine()
    rng = np.random.default_rng(2)
    thetas = np.stack([comp.theta(draw_params("plpeak", rng)) for _ in range(12)])
    ll, grads, kms = eng.evaluate_sequence(thetas, total, min_neff_cut=False, timing_every=5)
    for i, th in enumerate(thetas):
        r = eng.evaluate(th, total, min_neff_cut=False)
        assert r.log_likelihood == ll[i] and _same_grad(r.grad, grads[i], replay)
    assert np.all(kms[[0, 5, 10], 0] > 0) and np.all(kms[[1, 2, 3, 4, 6, 11]] == -1)
    ll2, grads2 = eng.evaluate_sequence(thetas, total, min_neff_cut=False)
    assert np.array_equal(ll, ll2) and _same_grad(grads, grads2, replay)
    with pytest.raises(ValueError):
        eng.evaluate_sequence(thetas[:, :-1], total)
    eng.close()


@pytest.mark.parametrize("replay", [False, True])
def test_aql_dispatch_path_equals_the_hip_stream_path(replay, monkeypatch):
    """Plain evaluations go through the engine's own AQL queue (gwinferno_amd/csrc/gwi_aql.h); timed ones, log-weights
    and batches through the HIP stream -- same kernels, other queue: identical bits, in any interleaving, also from
    several engines (= several producers into the shared queue pool) at once."""
    if replay:
        monkeypatch.setenv("GWI_DETERMINISTIC", "1")
    import threading

    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(9, 300, 3000, seed=21)
    comps = [COMPOSITIONS["bspline_test"](pe, inj) for _ in range(6)]
    engs = [c.engine() for c in comps]
    eng = engs[0]
    if "not found" in eng.dispatch_info() or "disabled by GWI_AQL=0" in eng.dispatch_info():  # no raw code object next to the library, or the suite is being run on the HIP stream on purpose
        pytest.skip(eng.dispatch_info())
    assert eng.dispatch_info() == "aql: active", eng.dispatch_info()
    rng = np.random.default_rng(8)
    thetas = np.stack([comps[0].theta(draw_params("bspline_test", rng)) for _ in range(10)])
    aql = [eng.evaluate(t, total, min_neff_cut=False) for t in thetas]
    lpe, linj = eng.log_weights(thetas[0])  # HIP stream (runs one evaluation through the AQL queue first)
    eng.set_timing(2)  # timed through the HIP stream
    hip = [eng.evaluate(t, total, min_neff_cut=False) for t in thetas]
    ms_hip = eng.last_kernel_ms()
    eng.set_timing(1)  # timed on the AQL queue (dispatch timestamps)
    aql_timed = [eng.evaluate(t, total, min_neff_cut=False) for t in thetas]
    ms_aql = eng.last_kernel_ms()
    eng.set_timing(0)
    for a, b, c in zip(aql, hip, aql_timed):
        assert a.log_likelihood == b.log_likelihood and _same_grad(a.grad, b.grad, replay) and np.array_equal(a.log_bfs, b.log_bfs) and np.array_equal(a.norms, b.norms)
        assert a.log_likelihood == c.log_likelihood and _same_grad(a.grad, c.grad, replay)
    # the two clocks bracket the same kernels: a few microseconds each, within a factor of two of one another
    assert 1e-3 < ms_hip[0] < 0.1 and 1e-3 < ms_aql[0] < 0.1 and 0.5 < ms_aql[0] / ms_hip[0] < 2.0 and ms_aql[1] > 1e-3
    batch = eng.evaluate_batch(thetas[:5], total, min_neff_cut=False)  # HIP stream again
    again = eng.evaluate(thetas[3], total, min_neff_cut=False)          # and back
    assert again.log_likelihood == aql[3].log_likelihood and rel_err(batch[3].log_likelihood, aql[3].log_likelihood) < 1e-12
    assert np.isfinite(lpe).any() and np.isfinite(linj).any()
    # six engines hammering the pool from six threads
    want = np.array([r.log_likelihood for r in aql] * 40)
    seq = np.concatenate([thetas] * 40)
    bad = [None] * len(engs)

    def work(k):
        ll, _ = engs[k].evaluate_sequence(seq, total, min_neff_cut=False)
        bad[k] = int(np.sum(ll != want))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(len(engs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert bad == [0] * len(engs)
    for e in engs:
        e.close()


def test_pinning_a_thread_next_to_the_gpu():
    """gwi_pin_thread_to_engine / _to_device: the calling thread's affinity shrinks to (a subset of) what it had, and
    evaluations from the pinned thread return the same bits.  Done in a worker thread: the test runner's own affinity
    stays as it was."""
    import os
    import threading

    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.engine import pin_thread_to_device
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(5, 200, 2000, seed=3)
    comp = COMPOSITIONS["plpeak"](pe, inj)
    eng = comp.engine()
    th = comp.theta(draw_params("plpeak", np.random.default_rng(1)))
    ref = eng.evaluate(th, total, min_neff_cut=False)
    out = {}

    def work():
        before = os.sched_getaffinity(0)
        out["changed"] = eng.pin_thread()
        out["device"] = pin_thread_to_device(0)
        after = os.sched_getaffinity(0)
        out["subset"] = after <= before and len(after) >= 1
        out["ll"] = eng.evaluate(th, total, min_neff_cut=False).log_likelihood

    before_main = os.sched_getaffinity(0)
    t = threading.Thread(target=work)
    t.start()
    t.join()
    assert out["subset"] and out["ll"] == ref.log_likelihood and out["changed"] == out["device"]
    assert os.sched_getaffinity(0) == before_main
    eng.close()


@pytest.mark.parametrize("env", [
    {"GWI_AQL_TAIL": "0"},        # the scan's whole argument block into a ring slot per launch
    {"GWI_AQL_HANDOFF": "hdp"},   # HDP flush register write + read-back instead of the kernel-argument read-back
    {"GWI_GACC_REP": "8"},        # replica counts other than 16 run the SAFE instantiation of spline models
    {"GWI_GACC_REP": "32"},
    {"GWI_AQL": "0"},             # everything on the HIP stream
], ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
@pytest.mark.parametrize("comp_name", ["plpeak", "bspline_iid"])
def test_launch_path_variants_agree(comp_name, env, monkeypatch):
    """The launch-path knobs (argument hand-off, replica count, dispatch path) must not change results: a run of different
    hyper-parameter points through an engine created under each knob equals the default engine's (values and scalar
    gradients bit for bit where the reduction order is the same, everything to 1e-12) and the C oracle's (1e-8 of the
    gradient's scale).  Consecutive DIFFERENT points are what would expose a stale kernel-argument slot."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle.c_oracle import COracle

    pe, inj, total = make_catalog(20, 700, 9000, seed=31)
    rng = np.random.default_rng(8)
    base = COMPOSITIONS[comp_name](pe, inj)
    eng0 = base.engine()
    thetas = [base.theta(draw_params(comp_name, rng)) for _ in range(12)]
    ref = [eng0.evaluate(t, total, min_neff_cut=False) for t in thetas]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    eng1 = COMPOSITIONS[comp_name](pe, inj).engine()
    if "GWI_AQL" in env:
        assert not eng1.dispatch_info().startswith("aql: active")
    orc = COracle(eng1.bound)
    for t, r0 in zip(thetas, ref):
        r1 = eng1.evaluate(t, total, min_neff_cut=False)
        same_order = "GWI_GACC_REP" not in env
        assert rel_err(r1.log_likelihood, r0.log_likelihood) < (1e-15 if same_order else 1e-12)
        assert np.allclose(r1.log_bfs, r0.log_bfs, rtol=1e-12, atol=0)
        assert np.allclose(r1.grad, r0.grad, rtol=1e-11, atol=1e-12 * max(1.0, float(np.max(np.abs(r0.grad)))))
        c = orc.evaluate(t, total, min_neff_cut=False)
        scale = max(1.0, float(np.max(np.abs(c["grad"]))))
        assert float(np.max(np.abs(r1.grad - c["grad"]))) / scale < 1e-8
        assert rel_err(r1.log_likelihood, c["log_likelihood"]) < VALUE_RTOL
    eng0.close()
    eng1.close()


@pytest.mark.parametrize("comp_name", ["pl_test", "plpeak_full", "plpeak_smooth", "plpeak_default_tilt", "bspline_iid", "bspline_full", "bspline_chieff", "bspline_redshift",
                                       "chm_powerlaw", "chm_bspline"])
def test_device_gradient_against_finite_differences_of_the_device_value(comp_name):
    """One cheap check per term kind that shares NO formula with the oracles: the engine's analytic gradient
    against fourth-order central differences of the engine's own log-likelihood along a few hyper-parameters.  (Together
    these compositions run every term kind: PL, PL+Peak, PL ratio incl. log m1 from the spline column, Beta, both tilt
    mixtures, PL z, exp- / linear / lerp splines, smooth, PL+Peak-smooth, PL with sampled bounds.)"""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog

    pe, inj, total = make_catalog(6, 400, 6000, seed=23)
    comp = COMPOSITIONS[comp_name](pe, inj)
    eng = comp.engine()
    rng = np.random.default_rng(4)
    th = comp.theta(draw_params(comp_name, rng))
    base = eng.evaluate(th, total, min_neff_cut=False)
    assert np.isfinite(base.log_likelihood)
    scale = max(1.0, float(np.max(np.abs(base.grad))))
    moving = np.flatnonzero(base.grad != 0.0)  # pinned slots / bounds of a sharp power law have zero gradient by construction
    for p in rng.choice(moving, size=min(5, moving.size), replace=False):
        h = 1e-4 * max(1.0, abs(th[p]))
        vals = []
        for k in (-2, -1, 1, 2):
            t = th.copy()
            t[p] += k * h
            vals.append(eng.evaluate(t, total, min_neff_cut=False, want_grad=False).log_likelihood)
        fd = (vals[0] - 8.0 * vals[1] + 8.0 * vals[2] - vals[3]) / (12.0 * h)
        assert abs(fd - base.grad[p]) < 2e-6 * scale, (comp_name, int(p), fd, float(base.grad[p]))
    eng.close()


@pytest.mark.parametrize("comp_name,n_ev,n_pe,n_inj,env", [
    ("plpeak", 11, 700, 5003, {"GWI_PBATCH_PTS": "4"}),              # ragged tiles; the BASELINE config-2 chain (a catalog this small would get one point per row: forced)
    ("plpeak_full", 7, 1000, 4000, {"GWI_PBATCH_PTS": "8"}),         # config 1: 16 scalar sums per sample -> two butterflies per point
    ("plpeak", 5, 300, 70, {"GWI_PBATCH_PTS": "3"}),                 # rows of 3, 3, ... points: a last row with fewer; waves without samples
    ("plpeak_smooth", 9, 512, 2048, {"GWI_PBATCH_PTS": "16"}),       # exactly one full trip per tile; one grid row for the whole batch
    ("chm_powerlaw", 6, 900, 3000, {"GWI_PBATCH_PTS": "5"}),         # theta-dependent truncation (POWERLAW_BOUNDS)
    ("plpeak", 20, 30000, 300000, {"GWI_PBATCH": "1"}),              # tiles of several trips for single evaluations: batches on single-trip tiles of their own; from 8 points on the BALANCED mode ((tile, point) units dealt out evenly)
    ("plpeak", 8, 800, 6000, {"GWI_MAX_BATCH": "40", "GWI_PBATCH_PTS": "16"}),  # 40 points: grid rows of 16, 16 and 8 (the LDS staging holds 16)
    ("plpeak", 69, 5000, 50000, {"GWI_PBATCH": "1", "GWI_MAX_BATCH": "40"}),    # BASELINE config 2's shape: balanced mode, 788 tiles x 16 / 40 points over one round of workgroups (segments of up to 16 points)
    ("plpeak_full", 3, 130, 200, {"GWI_PBATCH": "1"}),               # balanced mode with fewer units than workgroup slots: one unit per workgroup
])
def test_one_load_per_sample_batches_of_parametric_models(comp_name, n_ev, n_pe, n_inj, env, monkeypatch):
    """scan_pbatch_kernel (batched launches of models without spline terms: every sample loaded once for the points a workgroup
    draws; GWI_PBATCH=1 or a row size -- the default since round 6 is the one-grid-row-per-point kernel) against the C oracle,
    against single evaluations and against that default, in rows mode and in balanced mode, for batch sizes on both sides of the
    host-final / device-final switch, with and without the squared-weight pass."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle.c_oracle import COracle

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pe, inj, total = make_catalog(n_ev, n_pe, n_inj, seed=77)
    comp = COMPOSITIONS[comp_name](pe, inj)
    eng = comp.engine()
    assert eng.batch_path(16) == "pbatch", eng.batch_path(16)
    for k in env:
        monkeypatch.delenv(k)
    if "GWI_MAX_BATCH" in env:
        monkeypatch.setenv("GWI_MAX_BATCH", env["GWI_MAX_BATCH"])
    old = COMPOSITIONS[comp_name](pe, inj).engine()  # the default: one grid row per point
    assert old.batch_path(16) == "rows-per-point"
    orc = COracle(eng.bound)
    rng = np.random.default_rng(15)

    def draw():
        p = draw_params(comp_name, rng)
        if comp_name == "chm_powerlaw":
            p["mmin"], p["mmax"] = rng.uniform(3.0, 8.0), rng.uniform(70.0, 120.0)
        return comp.theta(p)

    for K in (1, 2, 5, 16) + ((40,) if env.get("GWI_MAX_BATCH") == "40" else ()):
        thetas = np.stack([draw() for _ in range(K)])
        for flags in (dict(min_neff_cut=False), dict(min_neff_cut=False, marginalize_selection=True)):
            if flags.get("marginalize_selection") and K not in (2, 16, 40):
                continue
            new_b, old_b = eng.evaluate_batch(thetas, total, **flags), old.evaluate_batch(thetas, total, **flags)
            for k in range(K):
                one = eng.evaluate(thetas[k], total, **flags)
                b = new_b[k]
                assert rel_err(b.log_likelihood, one.log_likelihood) < 1e-12
                assert np.allclose(b.log_bfs, one.log_bfs, rtol=1e-12, atol=1e-12) and np.allclose(b.log_neffs, one.log_neffs, rtol=1e-10, atol=1e-12)
                assert np.allclose(b.grad, one.grad, rtol=1e-10, atol=1e-11)
                assert np.allclose(b.norms, one.norms, rtol=1e-13)
                assert rel_err(b.log_likelihood, old_b[k].log_likelihood) < 1e-12 and np.allclose(b.grad, old_b[k].grad, rtol=1e-10, atol=1e-11)
                if k in (0, K - 1):
                    ref = orc.evaluate(thetas[k], total, **flags)
                    assert rel_err(b.log_likelihood, ref["log_likelihood"]) < VALUE_RTOL
                    scale = max(1.0, float(np.max(np.abs(ref["grad"]))))
                    assert float(np.max(np.abs(b.grad - ref["grad"]))) / scale < 1e-8
    # a batch repeats bit for bit (fixed summation order everywhere)
    thetas = np.stack([draw() for _ in range(16)])
    a, b = eng.evaluate_batch(thetas, total, min_neff_cut=False), eng.evaluate_batch(thetas, total, min_neff_cut=False)
    for x, y in zip(a, b):
        assert x.log_likelihood == y.log_likelihood and np.array_equal(x.grad, y.grad)
    eng.close()
    old.close()


def test_the_batched_kernel_of_a_spline_model_static_by_default_measured_on_request(monkeypatch):
    """Spline models have two batched kernels (matrix cores / 4-tap) that sum in different orders.  By DEFAULT the choice is a
    static rule (matrix cores from 9 points on, up to 8 gradient tiles): two handles of one model give the same bits, nothing is
    timed.  GWI_BATCH_AUTOTUNE=1 opts into the measurement: the engine times both on its first batched launch of >= 9 points and
    keeps the faster; gwi_batch_path then answers with that choice."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog

    for v in ("GWI_BATCH_MFMA", "GWI_BATCH_ROWS", "GWI_BATCH_AUTOTUNE"):
        monkeypatch.delenv(v, raising=False)
    pe, inj, total = make_catalog(11, 700, 5003, seed=41)
    comp = COMPOSITIONS["bspline_iid"](pe, inj)
    rng = np.random.default_rng(9)
    thetas = np.stack([comp.theta(draw_params("bspline_iid", rng)) for _ in range(16)])
    fixed, twin = comp.engine(), COMPOSITIONS["bspline_iid"](pe, inj).engine()
    a, b = fixed.evaluate_batch(thetas, total, min_neff_cut=False), twin.evaluate_batch(thetas, total, min_neff_cut=False)
    for e in (fixed, twin):
        assert e.batch_path(16) == "mfma" and not e.batch_calibration()["measured"]
    for k in range(16):  # same model, same catalog, another handle: the same kernel, the same bits (the matrix-core gradient is fixed-order)
        assert a[k].log_likelihood == b[k].log_likelihood and np.array_equal(a[k].grad, b[k].grad)
    fixed.close()
    twin.close()
    monkeypatch.setenv("GWI_BATCH_AUTOTUNE", "1")
    eng = COMPOSITIONS["bspline_iid"](pe, inj).engine()
    assert eng.batch_path(16) == "mfma" and not eng.batch_calibration()["measured"]  # the static rule until something has been measured
    small = eng.evaluate_batch(thetas[:4], total, min_neff_cut=False)                # below 9 points: nothing to choose
    assert not eng.batch_calibration()["measured"]
    batch = eng.evaluate_batch(thetas, total, min_neff_cut=False)
    cal = eng.batch_calibration()
    assert cal["measured"] and cal["mfma_us"] > 0 and cal["taps_us"] > 0
    assert eng.batch_path(16) == ("mfma" if cal["mfma_us"] <= cal["taps_us"] else "taps")
    for k in range(16):
        one = eng.evaluate(thetas[k], total, min_neff_cut=False)
        assert rel_err(batch[k].log_likelihood, one.log_likelihood) < 1e-12 and np.allclose(batch[k].grad, one.grad, rtol=1e-10, atol=1e-11)
        if k < 4:
            assert rel_err(small[k].log_likelihood, one.log_likelihood) < 1e-12
    eng.close()
