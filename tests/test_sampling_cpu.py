"""CPU: the built-in samplers (gwinferno_amd/sampling.py) on a target with a known answer -- a correlated
Gaussian with scales spanning 50x -- so that the drivers used in the GPU end-to-end tests are themselves checked."""
import numpy as np
import pytest

from gwinferno_amd.sampling import hmc, nuts


def _gaussian(seed=0, dim=5):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(dim, dim))
    cov = a @ a.T + np.diag([0.01, 1.0, 25.0, 4.0, 0.25][:dim])
    prec = np.linalg.inv(cov)
    mean = np.array([1.0, -2.0, 3.0, 0.0, 10.0][:dim])

    def target(x):
        d = x - mean
        return -0.5 * d @ prec @ d, -prec @ d

    return target, mean, cov


@pytest.mark.parametrize("sampler,kw", [(nuts, dict(n_warmup=500, n_samples=2500)), (hmc, dict(n_warmup=500, n_samples=4000, n_leapfrog=12))])
def test_samplers_recover_a_gaussian(sampler, kw):
    target, mean, cov = _gaussian()
    out = sampler(target, np.zeros(5), seed=3, **kw)
    s = out["samples"]
    sd = np.sqrt(np.diag(cov))
    assert 0.55 < out["accept_rate"] <= 1.0
    assert np.all(np.abs(s.mean(0) - mean) < 0.2 * sd)
    assert np.all(np.abs(s.var(0) / np.diag(cov) - 1) < 0.25)
    assert np.max(np.abs(np.corrcoef(s.T) - cov / np.outer(sd, sd))) < 0.15


def test_nuts_is_reproducible_and_counts_evaluations():
    target, _, _ = _gaussian()
    calls = [0]

    def counted(x):
        calls[0] += 1
        return target(x)

    a = nuts(counted, np.zeros(5), n_warmup=50, n_samples=50, seed=9)
    assert a["n_evals"] == calls[0]
    b = nuts(target, np.zeros(5), n_warmup=50, n_samples=50, seed=9)
    assert np.array_equal(a["samples"], b["samples"])
    assert a["tree_depth"].max() <= 10 and a["samples"].shape == (50, 5)


def test_nuts_survives_a_wall():
    """A target that returns the engine's cut sentinel (-1.797e308, zero gradient) outside a box: such leaves
    are treated as divergent and never accepted."""
    def target(x):
        if np.any(np.abs(x) > 3.0):
            return -1.7976931348623157e308, np.zeros_like(x)
        return -0.5 * float(x @ x), -x

    out = nuts(target, np.zeros(3), n_warmup=200, n_samples=400, seed=2)
    assert np.all(np.abs(out["samples"]) <= 3.0) and np.all(np.isfinite(out["log_prob"]))
    assert np.all(np.abs(out["samples"].mean(0)) < 0.3)


def test_interleaved_chains_equal_separate_runs():
    """nuts_chains drives several generators at once through begin/end pairs; each chain must produce exactly what
    a separate nuts() run with its seed produces (the interleaving only reorders WHEN evaluations happen)."""
    from gwinferno_amd.sampling import nuts_chains

    target, _, _ = _gaussian()

    def pair():
        box = {}

        def begin(x):
            box["r"] = target(np.array(x))

        def end():
            return box.pop("r")

        return begin, end

    starts = [np.zeros(5), np.ones(5), -np.ones(5)]
    res = nuts_chains([pair() for _ in starts], starts, n_warmup=40, n_samples=30, seed=5)
    for c, r in enumerate(res):
        alone = nuts(target, starts[c], n_warmup=40, n_samples=30, seed=5 + 1000 * c)
        assert np.array_equal(r["samples"], alone["samples"]) and r["n_evals"] == alone["n_evals"]


# ---- the library's C++ sampler (include/gwi_sampler.h) through its callback entry; no GPU involved ----
def test_native_nuts_recovers_a_gaussian():
    from gwinferno_amd.sampling import nuts_native

    target, mean, cov = _gaussian()
    out = nuts_native(target, np.zeros(5), n_warmup=500, n_samples=2500, seed=3)
    s = out["samples"]
    sd = np.sqrt(np.diag(cov))
    assert 0.55 < out["accept_rate"] <= 1.0 and out["n_divergent"] == 0
    assert np.all(np.abs(s.mean(0) - mean) < 0.2 * sd)
    assert np.all(np.abs(s.var(0) / np.diag(cov) - 1) < 0.25)
    assert np.max(np.abs(np.corrcoef(s.T) - cov / np.outer(sd, sd))) < 0.15
    lp = np.array([target(x)[0] for x in s[:20]])
    assert np.allclose(lp, out["log_prob"][:20], rtol=1e-12)


def test_native_nuts_is_reproducible_counts_and_propagates_errors():
    from gwinferno_amd.sampling import nuts_native

    target, _, _ = _gaussian()
    calls = [0]

    def counted(x):
        calls[0] += 1
        return target(x)

    a = nuts_native(counted, np.zeros(5), n_warmup=50, n_samples=50, seed=9)
    assert a["n_evals"] == calls[0]
    b = nuts_native(target, np.zeros(5), n_warmup=50, n_samples=50, seed=9)
    assert np.array_equal(a["samples"], b["samples"])
    c = nuts_native(target, np.zeros(5), n_warmup=50, n_samples=50, seed=10)
    assert not np.array_equal(a["samples"], c["samples"])
    assert a["tree_depth"].max() <= 10 and a["samples"].shape == (50, 5)

    def broken(x):
        if calls[0] > 0:
            calls[0] = -10**9
            raise ZeroDivisionError("target failed")
        return target(x)

    with pytest.raises(ZeroDivisionError):
        nuts_native(broken, np.zeros(5), n_warmup=5, n_samples=5)


def test_lockstep_chains_draw_what_separate_chains_draw():
    """gwi_nuts_run_lockstep: K chains (each the unchanged NUTS on a stack of its own), ONE batched target call per leapfrog
    step of all of them.  The batching only changes WHEN a chain's evaluations happen: chain c must draw, bit for bit,
    what gwi_nuts_run draws with seed + 1000 c -- also when chains finish at different times (the batch shrinks), when one
    chain starts on a dead point, and when the target fails mid-run."""
    from gwinferno_amd.sampling import nuts_native, nuts_native_lockstep

    target, _, _ = _gaussian()
    sizes = []

    def batch(xs, ids):
        sizes.append(len(ids))
        assert len(set(ids.tolist())) == len(ids)
        vg = [target(x) for x in xs]
        return np.array([v for v, _ in vg]), np.stack([g for _, g in vg])

    starts = np.stack([np.zeros(5), np.ones(5), -np.ones(5), 0.5 * np.ones(5), np.linspace(-1, 1, 5), 2 * np.ones(5), -0.3 * np.ones(5)])
    res = nuts_native_lockstep(batch, starts, n_warmup=60, n_samples=40, seed=5)
    total = 0
    for c, r in enumerate(res):
        alone = nuts_native(target, starts[c], n_warmup=60, n_samples=40, seed=5 + 1000 * c)
        assert np.array_equal(r["samples"], alone["samples"]) and np.array_equal(r["tree_depth"], alone["tree_depth"])
        assert r["n_evals"] == alone["n_evals"] and r["step_size"] == alone["step_size"] and r["accept_rate"] == alone["accept_rate"]
        total += r["n_evals"]
    assert sum(sizes) == total and max(sizes) == len(starts) and min(sizes) < len(starts)  # full batches, then a shrinking tail
    assert sizes[0] == len(starts) and np.mean(sizes) > 0.8 * len(starts)

    # a chain whose starting point is dead: the others are not disturbed, the call reports it
    def walled(xs, ids):
        v, g = batch(xs, ids)
        return np.where(xs[:, 0] > 5.0, -np.inf, v), g

    with pytest.raises(ValueError, match="starting point"):
        nuts_native_lockstep(walled, np.stack([np.zeros(5), 9.0 * np.ones(5)]), n_warmup=5, n_samples=5)

    # a failing target unwinds every chain and the exception comes back through the C frames
    calls = [0]

    def broken(xs, ids):
        calls[0] += 1
        if calls[0] > 7:
            raise ZeroDivisionError("target failed")
        return batch(xs, ids)

    with pytest.raises(ZeroDivisionError):
        nuts_native_lockstep(broken, starts[:3], n_warmup=20, n_samples=5)
    assert calls[0] == 8  # no evaluation after the failure
    again = nuts_native_lockstep(batch, starts[:2], n_warmup=10, n_samples=5, seed=1)  # and the library is usable afterwards
    assert again[0]["samples"].shape == (5, 5)


def test_a_queue_of_chains_keeps_the_batches_full_and_draws_the_same():
    """gwi_nuts_run_queue: 11 chains over 3 slots.  A chain that has drawn its last sample hands its slot to the next chain
    waiting, so the batches stay full until the queue is empty (without the queue they shrink as quick chains finish); chain c
    draws, bit for bit, what gwi_nuts_run draws with seed + 1000 c, whenever and next to whichever chains it ran.  Chains of
    very different lengths (different warm-up trajectories from different starts), a dead starting point in the middle of the
    queue and a failing target are covered."""
    from gwinferno_amd.sampling import lockstep_stats, nuts_native, nuts_native_lockstep

    target, _, _ = _gaussian()
    sizes = []

    def batch(xs, ids):
        sizes.append(len(ids))
        assert len(set(ids.tolist())) == len(ids)
        vg = [target(x) for x in xs]
        return np.array([v for v, _ in vg]), np.stack([g for _, g in vg])

    rng = np.random.default_rng(12)
    starts = rng.normal(size=(11, 5)) * np.linspace(0.2, 3.0, 11)[:, None]
    res = nuts_native_lockstep(batch, starts, n_warmup=40, n_samples=25, seed=9, slots=3)
    evals = []
    for c, r in enumerate(res):
        alone = nuts_native(target, starts[c], n_warmup=40, n_samples=25, seed=9 + 1000 * c)
        assert np.array_equal(r["samples"], alone["samples"]) and np.array_equal(r["tree_depth"], alone["tree_depth"])
        assert r["n_evals"] == alone["n_evals"] and r["step_size"] == alone["step_size"]
        evals.append(r["n_evals"])
    assert sum(sizes) == sum(evals) and max(sizes) == 3
    # never more than three at a time, and full almost to the end: only the last chains run in a shrinking batch
    tail = sorted(evals)[-1]
    assert np.mean(sizes) > 2.6 and sum(1 for s in sizes if s < 3) <= tail
    assert lockstep_stats()["mean_points_per_batch"] == pytest.approx(np.mean(sizes))
    # every chain at once (slots = 0): the same draws again
    flat = nuts_native_lockstep(batch, starts, n_warmup=40, n_samples=25, seed=9)
    for a, b in zip(res, flat):
        assert np.array_equal(a["samples"], b["samples"])

    # a dead starting point in the queue: reported, the library stays usable
    def walled(xs, ids):
        v, g = batch(xs, ids)
        return np.where(xs[:, 0] > 5.0, -np.inf, v), g

    bad = starts.copy()
    bad[6] = 9.0
    with pytest.raises(ValueError, match="starting point"):
        nuts_native_lockstep(walled, bad, n_warmup=5, n_samples=5, slots=3)
    calls = [0]

    def broken(xs, ids):
        calls[0] += 1
        if calls[0] > 30:
            raise ZeroDivisionError("target failed")
        return batch(xs, ids)

    with pytest.raises(ZeroDivisionError):
        nuts_native_lockstep(broken, starts, n_warmup=20, n_samples=5, slots=3)
    assert calls[0] == 31
    assert nuts_native_lockstep(batch, starts[:4], n_warmup=5, n_samples=5, seed=2, slots=2)[3]["samples"].shape == (5, 5)


def test_native_nuts_survives_a_wall_and_matches_the_numpy_sampler_statistically():
    from gwinferno_amd.sampling import nuts_native

    def target(x):
        if np.any(np.abs(x) > 3.0):
            return -1.7976931348623157e308, np.zeros_like(x)
        return -0.5 * float(x @ x), -x

    out = nuts_native(target, np.zeros(3), n_warmup=200, n_samples=1500, seed=2)
    ref = nuts(target, np.zeros(3), n_warmup=200, n_samples=1500, seed=2)
    assert np.all(np.abs(out["samples"]) <= 3.0) and np.all(np.isfinite(out["log_prob"]))
    assert np.all(np.abs(out["samples"].mean(0)) < 0.2)
    # same target, independent streams: second moments agree within Monte-Carlo error
    assert np.all(np.abs(out["samples"].var(0) - ref["samples"].var(0)) < 0.2)


def test_native_nuts_refuses_a_dead_starting_point():
    from gwinferno_amd.sampling import nuts_native

    def target(x):
        if x[0] > 1.0:
            return -1.7976931348623157e308, np.zeros_like(x)
        return -0.5 * float(x @ x), -x

    with pytest.raises(ValueError):
        nuts_native(target, np.array([2.0, 0.0]), n_warmup=5, n_samples=5)
    assert nuts_native(target, np.array([0.5, 0.0]), n_warmup=20, n_samples=20)["samples"].shape == (20, 2)


def test_find_map_climbs_to_the_mode_and_backs_off_a_wall():
    from gwinferno_amd.sampling import find_map

    target, mean, cov = _gaussian()
    out = find_map(target, np.zeros(5), Niter=4000, lr=0.05)
    sd = np.sqrt(np.diag(cov))
    assert np.all(np.abs(out["x"] - mean) < 0.05 * sd) and out["log_prob"] > -1e-2 and out["trace"][-1] > out["trace"][0]

    def walled(x):  # the mode sits just inside a cut
        if x[0] > 1.0:
            return -1.7976931348623157e308, np.zeros_like(x)
        return -0.5 * float((x[0] - 1.5) ** 2 + x[1] ** 2), np.array([-(x[0] - 1.5), -x[1]])

    out = find_map(walled, np.array([0.0, 1.0]), Niter=600, lr=0.05)
    assert 0.9 < out["x"][0] <= 1.0 and abs(out["x"][1]) < 0.1 and out["log_prob"] > -0.14  # stops at the wall, short of the (excluded) mode
    with pytest.raises(ValueError):
        find_map(walled, np.array([2.0, 0.0]))


def test_effective_sample_size_of_known_processes():
    """i.i.d. draws: ESS ~ number of draws; AR(1) with coefficient phi: ESS ~ N (1 - phi) / (1 + phi); a pinned column: NaN."""
    from gwinferno_amd.sampling import effective_sample_size

    rng = np.random.default_rng(0)
    iid = effective_sample_size(rng.normal(size=(4, 1000, 2)))
    assert np.all(iid > 2500) and np.all(iid < 5500)
    phi = 0.9
    y = np.zeros((4, 4000, 1))
    e = rng.normal(size=y.shape)
    for t in range(1, 4000):
        y[:, t] = phi * y[:, t - 1] + e[:, t]
    want = 4 * 4000 * (1 - phi) / (1 + phi)
    assert abs(effective_sample_size(y)[0] / want - 1) < 0.25
    assert np.isnan(effective_sample_size(np.ones((2, 100, 1)))[0])


def test_warmup_schedule_is_stans():
    """1000 warm-up iterations: 75 fast, slow windows ending at 100, 150, 250, 450, 950, 50 fast (Stan's / NumPyro's schedule);
    short warm-ups shrink to 15 % / 75 % / 10 %; below 20 there is no metric adaptation."""
    from gwinferno_amd.sampling import warmup_schedule

    assert warmup_schedule(1000) == (75, [100, 150, 250, 450, 950])
    assert warmup_schedule(300) == (75, [100, 150, 250])
    start, ends = warmup_schedule(100)
    assert start == 15 and ends[-1] == 90 and all(b > a for a, b in zip(ends[:-1], ends[1:]))
    assert warmup_schedule(10) == (10, [])


@pytest.mark.parametrize("native", [False, True])
def test_windowed_adaptation_learns_an_anisotropic_metric(native):
    """A Gaussian with standard deviations from 0.01 to 100: after the windowed warm-up the trees are short (the metric has
    absorbed the scales) and the variances come out right -- with ONE metric update at two thirds of the warm-up the chains
    needed depth-10 trees here."""
    from gwinferno_amd.sampling import nuts, nuts_native

    sig = np.array([0.01, 0.1, 1.0, 10.0, 100.0])

    def target(x):
        return float(-0.5 * np.sum((x / sig) ** 2)), -x / sig**2

    run = nuts_native if native else nuts
    r = run(target, np.ones(5) * 0.001, n_warmup=400, n_samples=600, seed=3)
    assert r["n_divergent"] == 0
    assert np.mean(r["tree_depth"]) < 4.5, np.mean(r["tree_depth"])
    got = np.std(r["samples"], axis=0)
    assert np.all(np.abs(got / sig - 1.0) < 0.25), got / sig


def test_split_rhat_tells_separated_chains_from_mixed_ones():
    """split_rhat ~ 1 for chains drawing from one distribution, >> 1 for chains in different places whose own ESS is fine
    (the multi-chain ESS then collapses to about the chain count: what bench.py's native_nuts reports for such a run)."""
    from gwinferno_amd.sampling import effective_sample_size, split_rhat

    rng = np.random.default_rng(5)
    x = rng.standard_normal((4, 400, 3))
    x[:, :, 2] = 1.5  # a pinned column
    r = split_rhat(x)
    assert np.all(np.abs(r[:2] - 1.0) < 0.05) and np.isnan(r[2])
    y = x.copy()
    y[1::2, :, 0] += 30.0  # two of the four chains elsewhere in coordinate 0
    r = split_rhat(y)
    assert r[0] > 10.0 and abs(r[1] - 1.0) < 0.05
    ess = effective_sample_size(y)
    assert ess[0] < 10.0 and ess[1] > 800.0
    assert all(effective_sample_size(y[c])[0] > 200.0 for c in range(4))
