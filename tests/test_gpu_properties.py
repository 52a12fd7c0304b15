"""GPU (-m gpu): size-independent properties of the hot path at the FULL BASELINE sizes (where the
NumPy oracle is too slow to be the checker for every point):
  * permutation invariance: shuffling the PE samples inside each event and the injections changes nothing
    beyond summation-order rounding;
  * replication: duplicating every injection (and doubling total_inj) leaves mu, log_l and the gradient
    unchanged and doubles n_eff_inj;
  * event additivity: sum_logBFs of a catalog == sum over two disjoint halves of its events;
  * shard + combine == unsharded (the multi-GPU path minus the exchange), 3 uneven shards;
  * gradient == central finite differences of the engine's own value (consistency of the analytic
    gradient at full size), and an oracle spot check on config 2."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(cfg):
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_config_catalog

    comp_name = {"c2": "plpeak", "c3": "bspline_iid", "c1": "plpeak_full"}[cfg]
    pe, inj, total = make_config_catalog(cfg)
    p = draw_params(comp_name, np.random.default_rng(77))
    return comp_name, pe, inj, total, p, COMPOSITIONS


@pytest.mark.parametrize("cfg", ["c2", "c3"])
def test_permutation_invariance_full_size(cfg):
    comp_name, pe, inj, total, p, C = _setup(cfg)
    comp = C[comp_name](pe, inj)
    a = comp.engine().evaluate(comp.theta(p), total, min_neff_cut=False)
    rng = np.random.default_rng(5)
    perm_pe = np.argsort(rng.random(pe["mass_1"].shape), axis=1)
    perm_inj = rng.permutation(inj["mass_1"].shape[0])
    pe2 = {k: np.take_along_axis(v, perm_pe, axis=1) for k, v in pe.items()}
    inj2 = {k: v[perm_inj] for k, v in inj.items()}
    comp2 = C[comp_name](pe2, inj2)
    b = comp2.engine().evaluate(comp2.theta(p), total, min_neff_cut=False)
    assert abs(a.log_likelihood - b.log_likelihood) < 1e-10 * abs(a.log_likelihood)
    assert np.allclose(a.log_bfs, b.log_bfs, rtol=1e-12, atol=1e-11)
    assert np.allclose(a.log_neffs, b.log_neffs, rtol=1e-10)
    assert np.allclose(a.grad, b.grad, rtol=1e-9, atol=1e-9)


def test_injection_replication_full_size():
    comp_name, pe, inj, total, p, C = _setup("c2")
    comp = C[comp_name](pe, inj)
    a = comp.engine().evaluate(comp.theta(p), total, min_neff_cut=False)
    inj2 = {k: np.concatenate([v, v]) for k, v in inj.items()}
    comp2 = C[comp_name](pe, inj2)
    b = comp2.engine().evaluate(comp2.theta(p), 2 * total, min_neff_cut=False)
    assert abs(a.summary.log_det_eff - b.summary.log_det_eff) < 1e-12
    assert abs(a.log_likelihood - b.log_likelihood) < 1e-10 * abs(a.log_likelihood)
    assert abs((b.summary.log_nEff_inj - a.summary.log_nEff_inj) - np.log(2.0)) < 1e-9
    assert np.allclose(a.grad, b.grad, rtol=1e-10, atol=1e-10)


def test_event_additivity_and_sharding_full_size():
    from gwinferno_amd.engine import NativePopulationLikelihood

    comp_name, pe, inj, total, p, C = _setup("c2")
    comp = C[comp_name](pe, inj)
    full = comp.engine()
    th = comp.theta(p)
    a = full.evaluate(th, total, min_neff_cut=False)
    # disjoint halves of the events (models built from the halves see different global zmin/zmax only
    # through the injections, which are shared, so per-event sites must agree exactly enough)
    recs = []
    for r in range(3):
        e = NativePopulationLikelihood(comp.weights(p, True), comp.weights(p, False), comp.hypervolume(p), rank=r, world=3)
        rec, lb, ln, lv = e.eval_partial(th)
        recs.append(rec)
        lo, hi = e.event_range
        assert np.allclose(lb + (a.summary.log_norm_const - np.log(e.n_pe)), a.log_bfs[lo:hi], rtol=1e-12, atol=1e-11)
        last = e
    out = last.combine(np.stack(recs), total, nobs=69, min_neff_cut=False)
    assert abs(out.log_likelihood - a.log_likelihood) < 1e-11 * abs(a.log_likelihood)
    assert abs(out.summary.sum_logBFs - a.summary.sum_logBFs) < 1e-10 * abs(a.summary.sum_logBFs)
    assert np.allclose(out.grad, a.grad, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("cfg", ["c2", "c3"])
def test_gradient_vs_own_finite_differences_full_size(cfg):
    comp_name, pe, inj, total, p, C = _setup(cfg)
    comp = C[comp_name](pe, inj)
    eng = comp.engine()
    th = comp.theta(p)
    g = eng.evaluate(th, total, min_neff_cut=False).grad
    rng = np.random.default_rng(3)
    for idx in rng.choice(len(th), size=min(6, len(th)), replace=False):
        h = 1e-4 * max(1.0, abs(th[idx]))
        vals = []
        for k in (-2, -1, 1, 2):
            t2 = th.copy()
            t2[idx] += k * h
            vals.append(eng.evaluate(t2, total, min_neff_cut=False, want_grad=False).log_likelihood)
        fd = (vals[0] - 8 * vals[1] + 8 * vals[2] - vals[3]) / (12 * h)
        assert abs(fd - g[idx]) < 1e-6 * max(1.0, abs(g[idx])), (idx, fd, g[idx])


def test_oracle_spot_check_config2_full_size():
    from oracle import numpy_oracle as O

    comp_name, pe, inj, total, p, C = _setup("c2")
    comp = C[comp_name](pe, inj)
    res = comp.engine().evaluate(comp.theta(p), total)  # reference defaults: min_neff_cut=True
    ref = O.COMPOSITIONS[comp_name](pe, inj).evaluate(p, total)
    assert abs(res.log_likelihood - float(ref["log_likelihood"])) <= 1e-9 * abs(float(ref["log_likelihood"]))
    assert np.max(np.abs(res.log_bfs - ref["logBFs"])) < 1e-9
    assert abs(res.summary.log_nEff_inj - float(ref["log_nEff_inj"])) < 1e-9


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
def test_non_finite_hyper_parameter_takes_the_nan_branch(bad):
    """A non-finite hyper-parameter makes every weight NaN in the reference -> log_l = nan_to_num(-inf)
    (analysis.py:287-289); the gradient of that constant is zero.  Evaluations before and after are
    unaffected."""
    from gwinferno_amd.engine import NEG_BIG

    comp_name, pe, inj, total, p, C = _setup("c1")
    comp = C[comp_name](pe, inj)
    eng = comp.engine()
    th = comp.theta(p)
    good = eng.evaluate(th, total, min_neff_cut=False)
    th_bad = th.copy()
    th_bad[0] = bad
    r = eng.evaluate(th_bad, total, min_neff_cut=False)
    assert r.log_likelihood == NEG_BIG
    assert np.all(r.grad == 0.0)
    again = eng.evaluate(th, total, min_neff_cut=False)
    assert again.log_likelihood == good.log_likelihood
    assert np.array_equal(again.grad, good.grad)


@pytest.mark.parametrize("cfg,comp_name", [("c5", "bspline_full"), ("c3", "bspline_iid"), ("c2", "plpeak"), ("c2", "bspline_iid"), ("c2", "bspline_defaults")])
def test_full_size_against_c_oracle(cfg, comp_name):
    """The BASELINE catalogs at FULL size (config 5: 2.5 M samples, 9 columns, 109 hyper-parameters) against the
    C/OpenMP oracle on the host: value, every per-event site and the whole gradient.  The last two are the size north_star's last
    sentence names -- 69 events x 5000 PE x 50 k injections (config 2's catalog) -- under B-spline models: config 3's composition,
    and the reference's DEFAULT spline counts (pipeline/utils.py:29-39, 104-155: 50 / 30 / 16 / 16 / 16 / 16 / 20 bases = 164
    coefficients + lamb; what `setup_bspline_*` builds with no arguments); for these also a batch of 16 points (the kernel the
    static rule picks: 12 gradient tiles keep the defaults on the 4-tap kernel) against the single evaluations."""
    from golden_util import rel_err

    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_config_catalog
    from oracle.c_oracle import COracle

    pe, inj, total = make_config_catalog(cfg)
    comp = COMPOSITIONS[comp_name](pe, inj)
    eng = comp.engine()
    orc = COracle(eng.bound)
    rng = np.random.default_rng(1)
    for _ in range(2):
        th = comp.theta(draw_params(comp_name, rng))
        got = eng.evaluate(th, total, min_neff_cut=False)
        ref = orc.evaluate(th, total, min_neff_cut=False)
        assert rel_err(got.log_likelihood, ref["log_likelihood"]) < 1e-9
        assert rel_err(got.log_bfs, ref["logBFs"]) < 1e-9
        assert rel_err(got.log_neffs, ref["log_nEffs"]) < 1e-8
        assert abs(got.summary.log_det_eff - ref["summary"].log_det_eff) < 1e-9 * abs(ref["summary"].log_det_eff)
        scale = max(1.0, float(np.max(np.abs(ref["grad"]))))
        assert float(np.max(np.abs(got.grad - ref["grad"]))) / scale < 1e-8
    if cfg == "c2" and comp_name != "plpeak":
        assert eng.n_theta == (165 if comp_name == "bspline_defaults" else 64)
        thetas = np.stack([comp.theta(draw_params(comp_name, rng)) for _ in range(16)])
        batch = eng.evaluate_batch(thetas, total, min_neff_cut=False)
        assert eng.batch_path(16) == ("taps" if comp_name == "bspline_defaults" else "mfma")
        for k in (0, 7, 15):
            one = eng.evaluate(thetas[k], total, min_neff_cut=False)
            assert rel_err(batch[k].log_likelihood, one.log_likelihood) < 1e-12
            assert np.allclose(batch[k].grad, one.grad, rtol=1e-10, atol=1e-11)
    eng.close()


@pytest.mark.parametrize("cfg,comp_name,blocks", [("c3", "bspline_iid", [9, 9, 9, 9, 9, 8, 8, 8]), ("c5", "bspline_full", [25] * 8)])
def test_eight_shards_equal_the_unsharded_engine_and_the_c_oracle(cfg, comp_name, blocks):
    """BASELINE config 4 (and the config-5 form of it): the B-spline catalogs split over EIGHT ranks -- contiguous event
    blocks 9,9,9,9,9,8,8,8 / 8 x 25 and equal injection slices (pipeline/analysis.py:78-86, :126-134) -- each
    shard scanned by its own engine (here all on one GPU), records combined as every rank does after the exchange.
    Against the unsharded engine AND the C oracle on the whole catalog: value, every site, the whole gradient."""
    from golden_util import rel_err

    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.engine import NativePopulationLikelihood
    from gwinferno_amd.synthetic import make_config_catalog
    from oracle.c_oracle import COracle

    pe, inj, total = make_config_catalog(cfg)
    comp = COMPOSITIONS[comp_name](pe, inj)
    full = comp.engine()
    orc = COracle(full.bound)
    p = draw_params(comp_name, np.random.default_rng(11))
    th = comp.theta(p)
    a = full.evaluate(th, total)  # reference defaults (min_neff_cut=True)
    ref = orc.evaluate(th, total)
    world = 8
    shards = [NativePopulationLikelihood(comp.weights(p, True), comp.weights(p, False), comp.hypervolume(p), rank=r, world=world) for r in range(world)]
    assert [e.n_ev for e in shards] == blocks
    n_inj = inj["mass_1"].shape[0]
    assert [e.n_inj for e in shards] == [n_inj // world] * world
    recs, lbs, lns, lvs = [], [], [], []
    for e in shards:
        rec, lb, ln, lv = e.eval_partial(th)
        recs.append(rec), lbs.append(lb), lns.append(ln), lvs.append(lv)
    for e in shards:  # every rank assembles the same result from the gathered records
        out = e.combine(np.stack(recs), total, nobs=full.n_ev)
        for want in (a, None):
            w_ll = want.log_likelihood if want else ref["log_likelihood"]
            w_grad = want.grad if want else ref["grad"]
            w_s = want.summary if want else ref["summary"]
            assert rel_err(out.log_likelihood, w_ll) < 1e-10
            assert rel_err(out.summary.sum_logBFs, w_s.sum_logBFs) < 1e-10
            assert abs(out.summary.log_det_eff - w_s.log_det_eff) < 1e-9 * abs(w_s.log_det_eff)
            assert abs(out.summary.log_nEff_inj - w_s.log_nEff_inj) < 1e-9 * abs(w_s.log_nEff_inj)
            assert rel_err(out.summary.variance_log_likelihood, w_s.variance_log_likelihood) < 1e-8
            assert abs(out.summary.min_log_nEff - w_s.min_log_nEff) < 1e-9 * max(1.0, abs(w_s.min_log_nEff))
            scale = max(1.0, float(np.max(np.abs(w_grad))))
            assert float(np.max(np.abs(out.grad - w_grad))) / scale < 1e-8
    shift = a.summary.log_norm_const - np.log(full.n_pe)
    assert np.allclose(np.concatenate(lbs) + shift, a.log_bfs, rtol=1e-12, atol=1e-11)
    assert rel_err(np.concatenate(lbs) + shift, ref["logBFs"]) < 1e-9
    assert np.allclose(np.concatenate(lns), a.log_neffs, rtol=1e-10)
    assert np.allclose(np.concatenate(lvs), a.variances, rtol=1e-9, atol=1e-14)
    for e in shards:
        e.close()
    full.close()


@pytest.mark.parametrize("env", [{}, {"GWI_BATCH_GEOMETRY": "0"}], ids=["own-geometry", "single-geometry"])
def test_batched_launch_at_full_size_equals_single_evaluations(env, monkeypatch):
    """Config 2 at full size, where batched launches (K >= 4) run on a launch geometry of their own (two trips per
    workgroup, gwi_create): values, sites and gradients of 4- and 16-point batches equal the single evaluations' (the tile
    boundaries differ, so summation-order rounding only), with and without that geometry, interleaved with single
    evaluations (which must find their own geometry again) and with a log-weight launch in between."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_config_catalog

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    pe, inj, total = make_config_catalog("c2")
    comp = COMPOSITIONS["plpeak"](pe, inj)
    eng = comp.engine()
    rng = np.random.default_rng(5)
    thetas = np.stack([comp.theta(draw_params("plpeak", rng)) for _ in range(16)])
    single = [eng.evaluate(t, total, min_neff_cut=False) for t in thetas]
    for K in (16, 4):
        batch = eng.evaluate_batch(thetas[:K], total, min_neff_cut=False)
        lw_pe, _ = eng.log_weights(thetas[0])  # a launch on the single geometry between batched ones
        assert np.isfinite(lw_pe).any()
        again = eng.evaluate(thetas[1], total, min_neff_cut=False)
        assert again.log_likelihood == single[1].log_likelihood and np.array_equal(again.grad, single[1].grad)
        for k in range(K):
            assert abs(batch[k].log_likelihood - single[k].log_likelihood) <= 1e-12 * abs(single[k].log_likelihood)
            assert np.allclose(batch[k].log_bfs, single[k].log_bfs, rtol=1e-12, atol=0)
            assert np.allclose(batch[k].log_neffs, single[k].log_neffs, rtol=1e-11, atol=0)
            scale = max(1.0, float(np.max(np.abs(single[k].grad))))
            assert float(np.max(np.abs(batch[k].grad - single[k].grad))) / scale < 1e-12
    eng.close()


def test_few_events_with_many_samples_get_at_most_16_tiles_per_event(monkeypatch):
    """One rank's share of config 5 on 8 GPUs (25 events x 10 000 PE samples + 62 500 injections): the combine launch fetches 16
    tile records per memory round trip, so the launch geometry gives an event at most 16 tiles where that leaves every CU a
    workgroup -- and the likelihood does not depend on the tiling (same values with the rule switched off: 40 tiles of 256)."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_config_catalog

    pe, inj, total = make_config_catalog("c5")
    comp = COMPOSITIONS["bspline_full"](pe, inj)
    eng = comp.engine(rank=0, world=8)
    geo = eng.launch_geometry()
    assert eng.n_ev == 25 and geo["tiles_per_event"] <= 16 and geo["n_scan_blocks"] >= 256, geo
    monkeypatch.setenv("GWI_TILE_CAP", "0")
    ref = COMPOSITIONS["bspline_full"](pe, inj).engine(rank=0, world=8)
    assert ref.launch_geometry()["tiles_per_event"] > 16
    rng = np.random.default_rng(8)
    for _ in range(2):
        th = eng.bound.theta_of(comp.weights(draw_params("bspline_full", rng), True))
        a, b = eng.evaluate(th, total, min_neff_cut=False), ref.evaluate(th, total, min_neff_cut=False)
        assert abs(a.log_likelihood - b.log_likelihood) <= 1e-12 * abs(b.log_likelihood)
        assert np.allclose(a.log_bfs, b.log_bfs, rtol=0, atol=1e-11) and np.allclose(a.grad, b.grad, rtol=1e-10, atol=1e-10 * np.max(np.abs(b.grad)))
    eng.close()
    ref.close()


def test_tiles_beyond_32768_samples_against_the_c_oracle():
    """Two events of 2.3 M posterior samples each: 64 tiles per event need tiles of >= 35 938 samples, which the scan's preloaded
    geometry carries in units of 256 (ScanHead::chunks, gwi_device.h) -- the launch geometry rounds them to whole multiples, and
    the likelihood, the per-event sites and the gradient still match the C oracle.  (Also: injections that start on an odd
    offset inside the columns' joint allocations -- 2 x 2 300 001 samples -- i.e. the padding of inj_offset().)"""
    from golden_util import rel_err

    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_catalog
    from oracle.c_oracle import COracle

    pe, inj, total = make_catalog(2, 2_300_001, 30_000, seed=41)
    comp = COMPOSITIONS["pl_test"](pe, inj)
    eng = comp.engine()
    geo = eng.launch_geometry()
    assert geo["chunk_pe"] >= 32768 and geo["chunk_pe"] % 256 == 0 and geo["tiles_per_event"] <= 64, geo
    orc = COracle(eng.bound)
    th = comp.theta(draw_params("pl_test", np.random.default_rng(4)))
    got = eng.evaluate(th, total, min_neff_cut=False)
    ref = orc.evaluate(th, total, min_neff_cut=False)
    assert rel_err(got.log_likelihood, ref["log_likelihood"]) < 1e-9
    assert rel_err(got.log_bfs, ref["logBFs"]) < 1e-9
    scale = max(1.0, float(np.max(np.abs(ref["grad"]))))
    assert float(np.max(np.abs(got.grad - ref["grad"]))) / scale < 1e-8
    eng.close()


def test_geometry_flips_between_single_and_batched_launches_many_times():
    """The scan's tile geometry travels as PRELOADED kernel arguments (ScanHead: the command processor reads them out of the
    argument block before the wave starts), and a batched launch (K >= 4) of config 2 runs on another geometry than a single
    evaluation, in the same persistent argument slots: 600 alternations single / batched / single must each reproduce the
    first results to the bit -- a stale head (the other launch's tile size or event count) would show as a different sum."""
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.synthetic import make_config_catalog

    pe, inj, total = make_config_catalog("c2")
    comp = COMPOSITIONS["plpeak"](pe, inj)
    eng = comp.engine()
    rng = np.random.default_rng(11)
    thetas = np.stack([comp.theta(draw_params("plpeak", rng)) for _ in range(4)])
    single0 = [eng.evaluate(t, total, min_neff_cut=False) for t in thetas]
    batch0 = eng.evaluate_batch(thetas, total, min_neff_cut=False)
    for it in range(600):
        k = it & 3
        s = eng.evaluate(thetas[k], total, min_neff_cut=False)
        assert s.log_likelihood == single0[k].log_likelihood and np.array_equal(s.grad, single0[k].grad), it
        b = eng.evaluate_batch(thetas, total, min_neff_cut=False)
        assert all(b[j].log_likelihood == batch0[j].log_likelihood and np.array_equal(b[j].grad, batch0[j].grad) for j in range(4)), it
    eng.close()
