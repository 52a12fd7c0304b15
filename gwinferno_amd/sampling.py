"""A small Hamiltonian Monte Carlo driver for environments without jax/numpyro.

The reference drives its model with ``numpyro.infer.NUTS`` (examples/utils.py:63-85), which needs
``jit(value_and_grad(potential_fn))``.  Where numpyro is installed the drop-in
``gwinferno_amd.likelihood.hierarchical_likelihood`` serves that unchanged; this module exists so that
the engine can be exercised end to end -- prior + likelihood + sampler -- with NumPy only.  It is a plain
HMC with dual-averaging step-size adaptation and a diagonal mass matrix estimated during warm-up; every
leapfrog step costs exactly one engine evaluation (value + gradient), as in NUTS.

``log_prob_and_grad(theta) -> (float, ndarray)`` is the full target: engine log-likelihood plus the
user's log-prior, e.g. the Normal priors and P-spline smoothing priors of
gwinferno/pipeline/utils.py:163-216.
"""
import numpy as np

from .smoothing import apply_difference_prior


class GaussianSmoothingPrior:
    """Independent Normal(0, sigma) priors on named slices of theta plus optional P-spline difference
    penalties (pipeline/utils.py:163-216): ``-0.5 * tau * ||D^deg c||^2``."""

    def __init__(self, n_theta):
        self.n_theta = n_theta
        self.sigmas = np.full(n_theta, np.inf)
        self.penalties = []  # (slice, tau, degree)

    def normal(self, sl, sigma):
        self.sigmas[sl] = sigma
        return self

    def smoothing(self, sl, tau, degree):
        self.penalties.append((sl, float(tau), int(degree)))
        return self

    def __call__(self, theta):
        with np.errstate(divide="ignore"):
            inv_var = np.where(np.isfinite(self.sigmas), 1.0 / self.sigmas**2, 0.0)
        lp = -0.5 * np.sum(inv_var * theta**2)
        grad = -inv_var * theta
        for sl, tau, deg in self.penalties:
            c = theta[sl]
            lp += apply_difference_prior(c, tau, deg)
            d = c
            for _ in range(deg):
                d = d[1:] - d[:-1]
            # gradient of -0.5 tau ||D^deg c||^2 = -tau (D^deg)^T (D^deg c)
            g = d
            for _ in range(deg):
                g = np.concatenate([[-g[0]], g[:-1] - g[1:], [g[-1]]])
            grad[sl] -= tau * g
        return lp, grad


class Bijector:
    """Unconstrained u -> constrained theta, elementwise, with log|d theta/d u| (what numpyro's
    ``biject_to(support)`` does for Uniform / HalfNormal / positive sites, e.g.
    examples/simple_powerlaw_peak_example.py:52-77)."""

    def __init__(self, n_theta):
        self.kind = np.zeros(n_theta, dtype=int)  # 0 identity, 1 interval(lo, hi), 2 positive
        self.lo = np.zeros(n_theta)
        self.hi = np.ones(n_theta)

    def interval(self, idx, lo, hi):
        self.kind[idx], self.lo[idx], self.hi[idx] = 1, lo, hi
        return self

    def positive(self, idx):
        self.kind[idx] = 2
        return self

    def fixed(self, idx, value):
        """theta[idx] is pinned to ``value`` (e.g. the first redshift-spline coefficient, pipeline/utils.py:213-214).
        The sampler's coordinate for it is an independent standard-normal dummy: log|J| stands in for its density."""
        self.kind[idx], self.lo[idx] = 3, value
        return self

    def forward(self, u):
        """theta, d theta/d u, d log|J| / d u, log|J|"""
        sig = 1.0 / (1.0 + np.exp(-u))
        width = self.hi - self.lo
        k = self.kind
        theta = np.where(k == 1, self.lo + width * sig, np.where(k == 2, np.exp(u), np.where(k == 3, self.lo, u)))
        dth = np.where(k == 1, width * sig * (1 - sig), np.where(k == 2, np.exp(u), np.where(k == 3, 0.0, 1.0)))
        with np.errstate(divide="ignore"):
            logj = np.where(k == 1, np.log(width * sig * (1 - sig)), np.where(k == 2, u, np.where(k == 3, -0.5 * u * u, 0.0)))
        dlogj = np.where(k == 1, 1 - 2 * sig, np.where(k == 2, 1.0, np.where(k == 3, -u, 0.0)))
        return theta, dth, dlogj, float(np.sum(logj))

    def inverse(self, theta):
        with np.errstate(all="ignore"):
            x = (theta - self.lo) / (self.hi - self.lo)
            return np.where(self.kind == 1, np.log(x / (1 - x)), np.where(self.kind == 2, np.log(theta), np.where(self.kind == 3, 0.0, theta)))


def make_target(engine, total_inj, prior, bijector=None, **likelihood_flags):
    """u -> (log posterior, gradient) in the sampler's (unconstrained) coordinates, using the engine's
    lean value_and_grad entry; ``prior`` acts on the constrained theta."""
    vg = engine.configure(total_inj, **likelihood_flags)

    def target(u):
        if bijector is None:
            theta, dth, dlogj, logj = u, 1.0, 0.0, 0.0
        else:
            theta, dth, dlogj, logj = bijector.forward(u)
        ll, g = vg(theta)
        lp, gp = prior(theta)
        return ll + lp + logj, (g + gp) * dth + dlogj

    return target


def hmc(target, theta0, n_warmup=200, n_samples=200, n_leapfrog=8, target_accept=0.8, seed=0, progress=None):
    """Returns dict(samples, log_prob, accept_rate, step_size, n_evals)."""
    rng = np.random.default_rng(seed)
    theta = np.array(theta0, dtype=np.float64)
    dim = theta.size
    lp, grad = target(theta)
    n_evals = 1
    inv_mass = np.ones(dim)
    # dual averaging (Hoffman & Gelman 2014, algorithm 5)
    eps = 0.01
    mu, log_eps_bar, h_bar, gamma, t0, kappa = np.log(10 * eps), 0.0, 0.0, 0.05, 10.0, 0.75
    samples, lps, accepts = [], [], []
    warm = []
    for it in range(n_warmup + n_samples):
        p0 = rng.normal(size=dim) / np.sqrt(inv_mass)
        th, g, p = theta.copy(), grad.copy(), p0.copy()
        cur_lp = lp
        h0 = -cur_lp + 0.5 * np.sum(inv_mass * p0**2)
        ok = True
        for _ in range(n_leapfrog):
            p = p + 0.5 * eps * g
            th = th + eps * inv_mass * p
            new_lp, g = target(th)
            n_evals += 1
            if not np.isfinite(new_lp) or new_lp < -1e300 or not np.all(np.isfinite(g)):
                ok = False
                break
            p = p + 0.5 * eps * g
        if ok:
            h1 = -new_lp + 0.5 * np.sum(inv_mass * p**2)
            acc = float(min(1.0, np.exp(min(0.0, h0 - h1))))
        else:
            acc = 0.0
        if rng.uniform() < acc:
            theta, lp, grad = th, new_lp, g
        if it < n_warmup:
            m = it + 1
            h_bar = (1 - 1 / (m + t0)) * h_bar + (target_accept - acc) / (m + t0)
            log_eps = mu - np.sqrt(m) / gamma * h_bar
            log_eps_bar = m**-kappa * log_eps + (1 - m**-kappa) * log_eps_bar
            eps = float(np.exp(log_eps))
            warm.append(theta.copy())
            if m == n_warmup // 2 and len(warm) > 20:  # one diagonal mass-matrix update mid warm-up
                var = np.var(np.array(warm[len(warm) // 2 :]), axis=0)
                inv_mass = np.where(var > 1e-12, var, 1.0)
            if m == n_warmup:
                eps = float(np.exp(log_eps_bar))
        else:
            samples.append(theta.copy())
            lps.append(lp)
            accepts.append(acc)
        if progress and (it + 1) % progress == 0:
            print(f"[hmc] iter {it + 1}: log_prob {lp:.3f} step {eps:.4g}", flush=True)
    return {"samples": np.array(samples), "log_prob": np.array(lps), "accept_rate": float(np.mean(accepts)) if accepts else 0.0, "step_size": eps, "n_evals": n_evals}


# ------------------------------------------------------------------------------------------------
# No-U-Turn sampler (the reference's sampler is numpyro.infer.NUTS, examples/utils.py:63-85)
#
# The sampler is written as a generator that YIELDS every point it needs evaluated and is SENT back
# (log_prob, gradient): `nuts` drives one such generator synchronously; `nuts_chains` drives several at once
# and keeps their engine evaluations in flight together (begin/end), which is how independent chains -- whose
# trajectories have different lengths and cannot be batched in lock step -- share one GPU.
# ------------------------------------------------------------------------------------------------
def _leapfrog(th, p, g, eps, inv_mass):
    p = p + 0.5 * eps * g
    th = th + eps * inv_mass * p
    lp, g = yield th
    p = p + 0.5 * eps * g
    return th, p, g, lp


def _find_step_size(th, lp, g, inv_mass, rng, eps=0.1):
    """Heuristic of Hoffman & Gelman (2014), algorithm 4: double / halve until the one-step acceptance
    probability crosses 1/2."""
    p = rng.normal(size=th.size) / np.sqrt(inv_mass)
    h0 = -lp + 0.5 * np.sum(inv_mass * p**2)

    def log_ratio(e):
        th1, p1, _, lp1 = yield from _leapfrog(th, p, g, e, inv_mass)
        h1 = -lp1 + 0.5 * np.sum(inv_mass * p1**2)
        return h0 - h1 if np.isfinite(h1) and lp1 > -1e300 else -np.inf

    a = 1.0 if (yield from log_ratio(eps)) > np.log(0.5) else -1.0
    for _ in range(50):
        if not a * (yield from log_ratio(eps)) > -a * np.log(2.0):
            break
        eps *= 2.0**a
    return eps


def warmup_schedule(n_warmup):
    """``(slow_start, window_ends)`` of the warm-up: Stan's schedule as NumPyro builds it
    (``numpyro.infer.hmc_util.build_adaptation_schedule``) -- 75 fast iterations (step size only), slow windows of 25, 50, 100, ...
    iterations (the last stretched to the end of the slow phase) each ending in a new diagonal metric, 50 fast iterations; shrunk to
    15 % / 75 % / 10 % below 150 warm-up iterations, no metric adaptation below 20.  ``gwi_sampler.cpp`` holds the same function."""
    n = int(n_warmup)
    if n < 20:
        return n, []
    init, term, base = 75, 50, 25
    if init + base + term > n:
        init, term = int(0.15 * n), int(0.1 * n)
        base = n - init - term
    slow_end, start, size, ends = n - term, init, base, []
    while start < slow_end:
        end = start + size
        if end + 2 * size > slow_end:
            end = slow_end
        ends.append(end)
        start, size = end, 2 * size
    return init, ends


def _nuts_gen(theta0, n_warmup, n_samples, max_tree_depth, target_accept, seed, progress, tag=""):
    """Generator: yields points to evaluate, receives (log_prob, grad); returns the result dict."""
    rng = np.random.default_rng(seed)
    theta = np.array(theta0, dtype=np.float64)
    dim = theta.size
    n_evals = [0]

    def evaluate(x):
        n_evals[0] += 1
        return (yield x)

    def leap(th, p, g, eps_):
        n_evals[0] += 1
        return (yield from _leapfrog(th, p, g, eps_, inv_mass))

    lp, grad = yield from evaluate(theta)
    inv_mass = np.ones(dim)

    def step_size(th, lp_, g_, eps0):
        gen = _find_step_size(th, lp_, g_, inv_mass, rng, eps=eps0)
        try:
            x = next(gen)
            while True:
                n_evals[0] += 1
                x = gen.send((yield x))
        except StopIteration as stop:
            return stop.value

    eps = yield from step_size(theta, lp, grad, 0.1)
    mu, log_eps_bar, h_bar, gamma, t0, kappa = np.log(10 * eps), 0.0, 0.0, 0.05, 10.0, 0.75
    da_count = 0

    def build(th, p, g, lp_, direction, depth, eps_, h0):
        """Subtree of 2^depth leapfrog states grown from (th, p, g) in `direction`.  Returns the outer edge,
        a multinomially drawn proposal, log sum of weights, summed momentum, and termination flags."""
        if depth == 0:
            th1, p1, g1, lp1 = yield from leap(th, p, g, direction * eps_)
            h1 = -lp1 + 0.5 * np.sum(inv_mass * p1**2)
            ok = np.isfinite(h1) and lp1 > -1e300 and np.all(np.isfinite(g1))
            d = h0 - h1 if ok else -np.inf
            return dict(edge=(th1, p1, g1, lp1), prop=(th1, lp1, g1), logw=d, rho=p1.copy(), p_first=p1, turning=False, diverging=(not ok) or d < -1000.0,
                        sum_alpha=float(min(1.0, np.exp(min(0.0, d)))) if ok else 0.0, n_alpha=1)
        a = yield from build(th, p, g, lp_, direction, depth - 1, eps_, h0)
        if a["diverging"] or a["turning"]:
            return a
        b = yield from build(*a["edge"], direction, depth - 1, eps_, h0)
        logw = np.logaddexp(a["logw"], b["logw"])
        prop = a["prop"]
        if not (b["diverging"] or b["turning"]) and np.log(rng.uniform()) < b["logw"] - logw:  # uniform over the subtree
            prop = b["prop"]
        rho = a["rho"] + b["rho"]
        # generalised U-turn test on this subtree: summed momentum against the velocities at its two ends
        p_first, p_last = a["p_first"], b["edge"][1]
        turning = b["turning"] or (np.dot(rho, inv_mass * p_first) <= 0) or (np.dot(rho, inv_mass * p_last) <= 0)
        return dict(edge=b["edge"], prop=prop, logw=logw, rho=rho, p_first=p_first, turning=turning, diverging=b["diverging"],
                    sum_alpha=a["sum_alpha"] + b["sum_alpha"], n_alpha=a["n_alpha"] + b["n_alpha"])

    samples, lps, accepts, depths = [], [], [], []
    n_div = 0
    slow_start, window_ends = warmup_schedule(n_warmup)
    next_window, w_count, w_mean, w_m2 = 0, 0, np.zeros(dim), np.zeros(dim)  # Welford accumulators of the current slow window
    for it in range(n_warmup + n_samples):
        p0 = rng.normal(size=dim) / np.sqrt(inv_mass)
        h0 = -lp + 0.5 * np.sum(inv_mass * p0**2)
        left = right = (theta, p0, grad, lp)
        prop = (theta, lp, grad)
        logw, rho = 0.0, p0.copy()
        sum_alpha, n_alpha, depth, diverged = 0.0, 0, 0, False
        while depth < max_tree_depth:
            direction = 1 if rng.uniform() < 0.5 else -1
            sub = yield from build(*(right if direction == 1 else left), direction, depth, eps, h0)
            sum_alpha += sub["sum_alpha"]
            n_alpha += sub["n_alpha"]
            if sub["diverging"]:
                diverged = True
                break
            if sub["turning"]:
                break
            if np.log(rng.uniform()) < sub["logw"] - logw:  # biased progressive sampling across doublings
                prop = sub["prop"]
            logw = np.logaddexp(logw, sub["logw"])
            rho = rho + sub["rho"]
            if direction == 1:
                right = sub["edge"]
            else:
                left = sub["edge"]
            depth += 1
            if (np.dot(rho, inv_mass * left[1]) <= 0) or (np.dot(rho, inv_mass * right[1]) <= 0):
                break
        theta, lp, grad = prop
        acc = sum_alpha / max(n_alpha, 1)
        n_div += int(diverged and it >= n_warmup)  # post-warm-up transitions only, as numpyro reports them (and gwi_sampler.cpp counts them)
        if it < n_warmup:
            m = it + 1
            da_count += 1
            h_bar = (1 - 1 / (da_count + t0)) * h_bar + (target_accept - acc) / (da_count + t0)
            log_eps = mu - np.sqrt(da_count) / gamma * h_bar
            log_eps_bar = da_count**-kappa * log_eps + (1 - da_count**-kappa) * log_eps_bar
            eps = float(np.exp(log_eps))
            if it >= slow_start and next_window < len(window_ends):  # slow phase: this draw feeds the window's variance
                w_count += 1
                delta = theta - w_mean
                w_mean = w_mean + delta / w_count
                w_m2 = w_m2 + delta * (theta - w_mean)
                if m == window_ends[next_window]:  # end of a window: new metric, re-found step size, dual averaging restarted
                    if w_count > 1:
                        var = w_m2 / (w_count - 1)
                        reg = (w_count / (w_count + 5.0)) * var + 1e-3 * (5.0 / (w_count + 5.0))  # Stan's shrinkage towards the unit metric
                        inv_mass = np.where(np.isfinite(reg) & (reg > 0), reg, 1.0)
                    w_count, w_mean, w_m2 = 0, np.zeros(dim), np.zeros(dim)
                    next_window += 1
                    eps = yield from step_size(theta, lp, grad, float(np.exp(log_eps_bar)))
                    mu, log_eps_bar, h_bar, da_count = np.log(10 * eps), 0.0, 0.0, 0
            if m == n_warmup:
                eps = float(np.exp(log_eps_bar)) if da_count > 0 else eps
        else:
            samples.append(theta.copy())
            lps.append(lp)
            accepts.append(acc)
            depths.append(depth)
        if progress and (it + 1) % progress == 0:
            print(f"[nuts{tag}] iter {it + 1}: log_prob {lp:.3f} step {eps:.4g} depth {depth}", flush=True)
    return {"samples": np.array(samples), "log_prob": np.array(lps), "accept_rate": float(np.mean(accepts)) if accepts else 0.0, "step_size": eps,
            "n_evals": n_evals[0], "tree_depth": np.array(depths), "n_divergent": n_div}


def nuts(target, theta0, n_warmup=200, n_samples=200, max_tree_depth=10, target_accept=0.8, seed=0, progress=None):
    """Multinomial NUTS with the generalised U-turn criterion (Betancourt 2017, as in Stan / NumPyro), dual-
    averaging step size and the windowed diagonal-metric warm-up of Stan / NumPyro (:func:`warmup_schedule`).  Every leapfrog step is one
    engine evaluation (value + gradient).  Returns dict(samples, log_prob, accept_rate, step_size,
    n_evals, tree_depth, n_divergent)."""
    gen = _nuts_gen(theta0, n_warmup, n_samples, max_tree_depth, target_accept, seed, progress)
    try:
        x = next(gen)
        while True:
            x = gen.send(target(x))
    except StopIteration as stop:
        return stop.value


def nuts_chains(begin_end, theta0s, post=None, n_warmup=200, n_samples=200, max_tree_depth=10, target_accept=0.8, seed=0, progress=None):
    """Independent NUTS chains with their evaluations in flight together.  ``begin_end[c] = (begin, end)`` of
    chain c's own engine (``NativePopulationLikelihood.configure_async``): ``begin(x)`` issues the evaluation of
    point x, ``end()`` returns ``(log_likelihood, grad)``.  ``post(x, ll, grad) -> (log_prob, grad)`` adds the
    prior / change of variables (default: identity); with a bijector pass ``pre`` inside it -- see
    :func:`make_async_target`.  Chains advance independently (different tree depths); while one chain's result is
    being turned into its next request, the others' kernels are running.  Returns a list of result dicts."""
    C = len(begin_end)
    gens = [_nuts_gen(theta0s[c], n_warmup, n_samples, max_tree_depth, target_accept, seed + 1000 * c, progress, tag=f" chain {c}") for c in range(C)]
    post = post or (lambda x, ll, g: (ll, g))
    pending = [None] * C
    results = [None] * C
    for c in range(C):
        pending[c] = next(gens[c])
        begin_end[c][0](pending[c])
    live = C
    while live:
        for c in range(C):
            if results[c] is not None:
                continue
            ll, g = begin_end[c][1]()
            try:
                pending[c] = gens[c].send(post(pending[c], ll, np.array(g)))
                begin_end[c][0](pending[c])
            except StopIteration as stop:
                results[c] = stop.value
                live -= 1
    return results


def make_async_target(engine, total_inj, prior, bijector=None, **likelihood_flags):
    """(begin, end) in the sampler's unconstrained coordinates for :func:`nuts_chains`: the counterpart of
    :func:`make_target` on the engine's begin/end entry."""
    eng_begin, eng_end = engine.configure_async(total_inj, **likelihood_flags)
    state = {}

    def begin(u):
        if bijector is None:
            theta, dth, dlogj, logj = u, 1.0, 0.0, 0.0
        else:
            theta, dth, dlogj, logj = bijector.forward(u)
        state["t"] = (theta, dth, dlogj, logj)
        eng_begin(theta)

    def end():
        ll, g = eng_end()
        theta, dth, dlogj, logj = state["t"]
        lp, gp = prior(theta)
        return ll + lp + logj, (g + gp) * dth + dlogj

    return begin, end


def find_map(target, theta0, Niter=100, lr=0.01, b1=0.9, b2=0.999, eps=1e-8):
    """MAP point of ``target(x) -> (log_prob, grad)`` by Adam ascent: what the reference's ``find_map``
    (analysis.py:24-47: SVI with an AutoDelta guide, ``numpyro.optim.Adam(lr)``, ``Niter`` steps) does in the
    sampler's unconstrained coordinates, started from ``theta0`` instead of the guide's random initial location.
    Steps that land on a cut (the ``nan_to_num(-inf)`` value, zero gradient) or on a non-finite value are undone
    and the step length halved.  Returns ``{"x", "log_prob", "trace"}`` with the best point visited."""
    x = np.array(theta0, dtype=np.float64)
    m, v = np.zeros_like(x), np.zeros_like(x)
    lp, g = target(x)
    if not (np.isfinite(lp) and lp > -1e300 and np.all(np.isfinite(g))):
        raise ValueError("find_map: the starting point has zero probability or a non-finite gradient")
    best = (float(lp), x.copy())
    trace = [float(lp)]
    for t in range(1, int(Niter) + 1):
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        step = lr * (m / (1 - b1**t)) / (np.sqrt(v / (1 - b2**t)) + eps)
        for _ in range(30):
            lp_new, g_new = target(x + step)
            if np.isfinite(lp_new) and lp_new > -1e300 and np.all(np.isfinite(g_new)):
                break
            step = 0.5 * step
        else:
            break
        x, lp, g = x + step, lp_new, g_new
        trace.append(float(lp))
        if lp > best[0]:
            best = (float(lp), x.copy())
    return {"x": best[1], "log_prob": best[0], "trace": np.array(trace)}


# ---- the library's native sampler (include/gwi_sampler.h) ----
def effective_sample_size(samples):
    """Bulk effective sample size per parameter of ``samples[chains, draws, dim]``: the multi-chain estimator with Geyer's
    initial monotone sequence (autocovariances by FFT; what ``arviz.ess`` / ``numpyro.diagnostics.effective_sample_size``
    compute, which ``mcmc.print_summary()`` shows in the reference's drivers, bin/gwinferno_run_from_config.py:70).  Constant
    columns (pinned parameters) get NaN."""
    x = np.asarray(samples, dtype=np.float64)
    if x.ndim == 2:
        x = x[None]
    C, n, dim = x.shape
    out = np.full(dim, np.nan)
    if n < 4:
        return out
    xc = x - x.mean(axis=1, keepdims=True)
    m = 1 << int(np.ceil(np.log2(2 * n)))
    f = np.fft.rfft(xc, n=m, axis=1)
    acov = np.fft.irfft(f * np.conj(f), n=m, axis=1)[:, :n, :] / n  # biased autocovariance per chain
    chain_var = acov[:, 0, :] * n / (n - 1.0)
    W = chain_var.mean(axis=0)
    B_over_n = x.mean(axis=1).var(axis=0, ddof=1) if C > 1 else 0.0
    var_plus = W * (n - 1.0) / n + B_over_n
    for j in range(dim):
        if not (var_plus[j] > 0.0) or not np.isfinite(var_plus[j]):
            continue
        rho = 1.0 - (W[j] - acov[:, :, j].mean(axis=0)) / var_plus[j]
        # Geyer: sums of adjacent pairs, truncated at the first negative pair, made monotone
        pairs = rho[: 2 * (n // 2)].reshape(-1, 2).sum(axis=1)
        neg = np.flatnonzero(pairs < 0.0)
        k = neg[0] if neg.size else len(pairs)
        p = np.minimum.accumulate(pairs[:k]) if k > 0 else np.zeros(0)
        tau = -1.0 + 2.0 * p.sum()
        tau = max(tau, 1.0 / np.log10(C * n))
        out[j] = C * n / tau
    return out


def split_rhat(samples):
    """Split R-hat per parameter of ``samples[chains, draws, dim]`` (each chain halved; Gelman et al. 2013, the ``r_hat``
    column of ``mcmc.print_summary()``).  Values far above 1 with healthy within-chain ESS mean the chains sit in different
    modes, not that they mix slowly.  Constant columns get NaN."""
    x = np.asarray(samples, dtype=np.float64)
    if x.ndim == 2:
        x = x[None]
    h = x.shape[1] // 2
    if h < 2:
        return np.full(x.shape[2], np.nan)
    y = np.concatenate([x[:, :h], x[:, h:2 * h]], axis=0)
    W = y.var(axis=1, ddof=1).mean(axis=0)
    B = h * y.mean(axis=1).var(axis=0, ddof=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        r = np.sqrt(((h - 1.0) / h * W + B / h) / W)
    return np.where(W > 0.0, r, np.nan)


def _nuts_options(n_warmup, n_samples, max_tree_depth, target_accept, seed):
    from . import _native as N

    return N.GwiNutsOptions(int(n_warmup), int(n_samples), int(max_tree_depth), 0, float(target_accept), int(seed))


def _result_dict(r):
    return dict(accept_rate=r.accept_rate, step_size=r.step_size, n_evals=int(r.n_evals), n_divergent=int(r.n_divergent))


def nuts_native(target, theta0, n_warmup=200, n_samples=200, max_tree_depth=10, target_accept=0.8, seed=0):
    """:func:`nuts` run by the engine library's C++ sampler (``gwi_nuts_run``) on a Python ``target(x) ->
    (log_prob, grad)``: the tree building happens natively, the target is called back per leapfrog step.
    Same return value as :func:`nuts`."""
    import ctypes as C

    from . import _native as N

    lib = N.load_library()
    theta0 = N.f64(theta0)
    d = theta0.size
    err = []

    def cb(_user, x, lp, grad):
        try:
            v, g = target(np.ctypeslib.as_array(x, (d,)).copy())
            lp[0] = v
            np.ctypeslib.as_array(grad, (d,))[:] = g
            return 0
        except Exception as exc:  # propagate through the C frames
            err.append(exc)
            return 1

    samples, logp, depth = np.empty((n_samples, d)), np.empty(n_samples), np.empty(n_samples, dtype=np.int32)
    res = N.GwiNutsResult()
    opt = _nuts_options(n_warmup, n_samples, max_tree_depth, target_accept, seed)
    st = lib.gwi_nuts_run(N.GWI_TARGET_FN(cb), None, d, N.as_dp(theta0), C.byref(opt), N.as_dp(samples), N.as_dp(logp), depth.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(res))
    if err:
        raise err[0]
    if st == -1:
        raise ValueError("gwi_nuts_run: the starting point has zero probability or a non-finite gradient")
    if st != 0:
        raise N.NativeEngineError(f"gwi_nuts_run: {N.STATUS_NAMES.get(st, st)}")
    out = _result_dict(res)
    out.update(samples=samples, log_prob=logp, tree_depth=depth)
    return out


def nuts_native_lockstep(batch_target, theta0s, n_warmup=200, n_samples=200, max_tree_depth=10, target_accept=0.8, seed=0, slots=0):
    """Lock-step chains on a Python BATCHED target (``gwi_nuts_run_queue``): ``batch_target(xs[k, d], chain_ids[k]) ->
    (log_probs[k], grads[k, d])`` is called once per leapfrog step of all running chains (numpyro's
    ``chain_method="vectorized"``).  Chain ``c`` uses seed ``seed + 1000 c`` and draws exactly what :func:`nuts_native`
    draws with that seed.  ``slots > 0``: at most that many chains run at a time; a chain that ends hands its slot to the next
    one waiting, so the batches stay full until the queue is empty.  Returns one dict per chain."""
    import ctypes as C

    from . import _native as N

    lib = N.load_library()
    theta0s = np.atleast_2d(N.f64(theta0s))
    n_chains, d = theta0s.shape
    err = []

    def cb(_user, k, ids, xs, lps, grads):
        try:
            v, g = batch_target(np.ctypeslib.as_array(xs, (k, d)).copy(), np.ctypeslib.as_array(ids, (k,)).copy())
            np.ctypeslib.as_array(lps, (k,))[:] = v
            np.ctypeslib.as_array(grads, (k, d))[:] = g
            return 0
        except Exception as exc:  # propagate through the C frames
            err.append(exc)
            return 1

    samples, logp, depth = np.empty((n_chains, n_samples, d)), np.empty((n_chains, n_samples)), np.empty((n_chains, n_samples), dtype=np.int32)
    res = (N.GwiNutsResult * n_chains)()
    opt = _nuts_options(n_warmup, n_samples, max_tree_depth, target_accept, seed)
    st = lib.gwi_nuts_run_queue(N.GWI_BATCH_TARGET_FN(cb), None, d, n_chains, int(slots), N.as_dp(theta0s), C.byref(opt), N.as_dp(samples), N.as_dp(logp),
                                depth.ctypes.data_as(C.POINTER(C.c_int32)), res)
    if err:
        raise err[0]
    if st == -1:
        raise ValueError("gwi_nuts_run_lockstep: a starting point has zero probability or a non-finite gradient")
    if st != 0:
        raise N.NativeEngineError(f"gwi_nuts_run_lockstep: {N.STATUS_NAMES.get(st, st)}")
    out = []
    for c in range(n_chains):
        r = _result_dict(res[c])
        r.update(samples=samples[c], log_prob=logp[c], tree_depth=depth[c])
        out.append(r)
    return out


def lockstep_stats():
    """Of this thread's last lock-step run (``gwi_nuts_lockstep_stats``): batches, mean points per batch, and where the wall
    time went per batch (collecting = mostly waiting for the GPU, the chains' own arithmetic, gathering + issuing)."""
    from . import _native as N

    out = np.zeros(6)
    N.load_library().gwi_nuts_lockstep_stats(N.as_dp(out))
    nb = max(out[0], 1.0)
    return {"batches": int(out[0]), "mean_points_per_batch": out[1] / nb, "collect_us_per_batch": 1e6 * out[2] / nb, "chains_us_per_batch": 1e6 * out[3] / nb,
            "issue_us_per_batch": 1e6 * out[4] / nb, "wall_s": out[5]}


def _engine_sampler_args(engines, n_theta, prior, bijector, theta0s, total_inj, likelihood_flags):
    from . import _native as N

    bij = bijector if bijector is not None else Bijector(n_theta)
    pri = (N.GwiParamPrior * n_theta)()
    for i in range(n_theta):
        pri[i] = N.GwiParamPrior(int(bij.kind[i]), 0, float(bij.lo[i]), float(bij.hi[i]), float(prior.sigmas[i]))
    pens = (N.GwiSmoothingPenalty * max(1, len(prior.penalties)))()
    for k, (sl, tau, deg) in enumerate(prior.penalties):
        start, stop, step = sl.indices(n_theta)
        if step != 1:
            raise ValueError("smoothing penalties act on contiguous slices")
        pens[k] = N.GwiSmoothingPenalty(start, stop - start, deg, 0, tau)
    u0 = N.f64(np.stack([bij.inverse(t) for t in theta0s]))
    if not np.all(np.isfinite(u0)):
        raise ValueError("starting points must lie strictly inside the parameter supports")
    return pri, pens, u0, engines[0]._options(total_inj, **likelihood_flags)


def nuts_engine_lockstep(engines, chains_per_engine, total_inj, prior, bijector, theta0s, n_warmup=200, n_samples=200, max_tree_depth=10, target_accept=0.8, seed=0,
                         **likelihood_flags):
    """Lock-step NUTS inside the library (``gwi_nuts_engine_lockstep``): ``len(engines) * chains_per_engine`` chains, the chains of
    group ``g`` on ``engines[g]``, every leapfrog step of a group ONE batched launch (``gwi_eval_batch``'s kernels) -- numpyro's
    ``chain_method="vectorized"`` (examples/utils.py:63-85).  One host thread; with two or three engines the groups' launches
    overlap each other's host arithmetic.  Same target and return value as :func:`nuts_engine` (``theta0s[c]``: constrained
    starting point of chain ``c``, chain ``c`` in group ``c // chains_per_engine``).  MORE starting points than
    ``len(engines) * chains_per_engine`` make a queue (``gwi_nuts_engine_queue``): every engine runs ``chains_per_engine`` chains
    at a time and a chain that ends hands its slot to the next one waiting, so the batches stay full; chain ``c`` draws what it
    would draw alone with seed ``seed + 1000 c`` wherever it runs."""
    import ctypes as C

    from . import _native as N

    engines = list(engines)
    theta0s = np.atleast_2d(N.f64(theta0s))
    n_chains, n_theta = theta0s.shape
    K = int(chains_per_engine)
    if n_chains < len(engines) * K:
        raise ValueError("theta0s must hold at least len(engines) * chains_per_engine starting points")
    lib = engines[0].lib
    pri, pens, u0, lopt = _engine_sampler_args(engines, n_theta, prior, bijector, theta0s, total_inj, likelihood_flags)
    handles = (C.c_void_p * len(engines))(*[e.handle for e in engines])
    samples = np.empty((n_chains, n_samples, n_theta))
    logp = np.empty((n_chains, n_samples))
    depth = np.empty((n_chains, n_samples), dtype=np.int32)
    res = (N.GwiNutsResult * n_chains)()
    opt = _nuts_options(n_warmup, n_samples, max_tree_depth, target_accept, seed)
    st = lib.gwi_nuts_engine_queue(handles, len(engines), K, n_chains, n_theta, C.byref(lopt), pri, pens, len(prior.penalties), N.as_dp(u0), C.byref(opt), N.as_dp(samples),
                                   N.as_dp(logp), depth.ctypes.data_as(C.POINTER(C.c_int32)), res)
    if st == -1:
        raise ValueError("gwi_nuts_engine_lockstep: a chain's starting point has zero likelihood (a cut, or outside the model's support) or a non-finite gradient")
    if st != 0:
        msgs = "; ".join(lib.gwi_last_error(e.handle).decode() for e in engines)
        raise N.NativeEngineError(f"gwi_nuts_engine_lockstep: {N.STATUS_NAMES.get(st, st)}: {msgs}")
    out = []
    for c in range(n_chains):
        r = _result_dict(res[c])
        r.update(samples=samples[c], log_prob=logp[c], tree_depth=depth[c])
        out.append(r)
    return out


def nuts_engine(engines, total_inj, prior, bijector, theta0s, n_warmup=200, n_samples=200, max_tree_depth=10, target_accept=0.8, seed=0, **likelihood_flags):
    """NUTS on ``engine log-likelihood + prior`` entirely inside the library (``gwi_nuts_engine``): one host
    thread and one engine per chain, no Python between two likelihood evaluations.

    ``engines`` is one :class:`~gwinferno_amd.engine.BoundEngine` per chain (each holds its own copy of the
    catalog and its own stream); ``prior`` a :class:`GaussianSmoothingPrior`, ``bijector`` a :class:`Bijector`
    (or ``None``); ``theta0s[c]`` the constrained starting point of chain ``c``.  Returns one dict per chain,
    as :func:`nuts` does, with ``samples`` in constrained coordinates."""
    import ctypes as C

    from . import _native as N

    engines = list(engines)
    theta0s = np.atleast_2d(N.f64(theta0s))
    n_chains, n_theta = theta0s.shape
    if len(engines) != n_chains:
        raise ValueError("one engine per chain")
    lib = engines[0].lib
    pri, pens, u0, lopt = _engine_sampler_args(engines, n_theta, prior, bijector, theta0s, total_inj, likelihood_flags)
    handles = (C.c_void_p * n_chains)(*[e.handle for e in engines])
    samples = np.empty((n_chains, n_samples, n_theta))
    logp = np.empty((n_chains, n_samples))
    depth = np.empty((n_chains, n_samples), dtype=np.int32)
    res = (N.GwiNutsResult * n_chains)()
    opt = _nuts_options(n_warmup, n_samples, max_tree_depth, target_accept, seed)
    st = lib.gwi_nuts_engine(handles, n_chains, n_theta, C.byref(lopt), pri, pens, len(prior.penalties), N.as_dp(u0), C.byref(opt), N.as_dp(samples), N.as_dp(logp),
                             depth.ctypes.data_as(C.POINTER(C.c_int32)), res)
    if st == -1:
        raise ValueError("gwi_nuts_engine: a chain's starting point has zero likelihood (a cut, or outside the model's support) or a non-finite gradient")
    if st != 0:
        msgs = "; ".join(lib.gwi_last_error(e.handle).decode() for e in engines)
        raise N.NativeEngineError(f"gwi_nuts_engine: {N.STATUS_NAMES.get(st, st)}: {msgs}")
    out = []
    for c in range(n_chains):
        r = _result_dict(res[c])
        r.update(samples=samples[c], log_prob=logp[c], tree_depth=depth[c])
        out.append(r)
    return out
