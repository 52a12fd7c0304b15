#!/usr/bin/env python3
"""End-to-end sampling with NumPy only: BASELINE config-2 model (PL+Peak m1 x PL q x PL z) on a
synthetic catalog, the priors of the reference's example, and the built-in NUTS driver (200 warm-up +
200 samples, as examples/simple_powerlaw_peak_example.py runs numpyro's NUTS; `--hmc` selects the
fixed-length HMC driver instead).  Every leapfrog step is one engine evaluation (value + gradient),
exactly what NUTS pays per step in the reference (examples/utils.py:63-85).
(Without the n_eff cuts a chain can fall into the known failure of importance-sampled likelihoods -- a
vanishing peak width with all weight on a few samples -- which is what `min_neff_cut` exists to forbid.)
With `--chains C` (C > 1) C independent NUTS chains run on the one GPU, one engine each, their evaluations in
flight together (gwi_eval_begin / gwi_eval_end): chains of different tree depths overlap instead of queueing.
`--native` runs the library's C++ sampler instead (gwi_nuts_engine, include/gwi_sampler.h): the same algorithm,
target and priors with no Python between two evaluations -- one host thread per chain, each on its own engine.
`--native --lockstep` advances the chains together instead (gwi_nuts_engine_lockstep: numpyro's chain_method="vectorized"):
every leapfrog step of up to 16 chains is ONE batched launch, two groups of chains alternate on one host thread
(`--chains 32 --native --lockstep`).
    python examples/sample_plpeak_hmc.py [n_events n_pe n_inj] [--hmc] [--neff-cut] [--chains C] [--native [--lockstep]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gwinferno_amd.compositions import COMPOSITIONS  # noqa: E402
from gwinferno_amd.sampling import (Bijector, GaussianSmoothingPrior, hmc, lockstep_stats, make_async_target, make_target, nuts, nuts_chains, nuts_engine,  # noqa: E402
                                    nuts_engine_lockstep)
from gwinferno_amd.synthetic import make_catalog  # noqa: E402

use_hmc = "--hmc" in sys.argv
n_chains = int(sys.argv[sys.argv.index("--chains") + 1]) if "--chains" in sys.argv else 1
argv = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and sys.argv[i - 1] != "--chains"]
n_ev, n_pe, n_inj = (int(x) for x in argv[:3]) if len(argv) >= 3 else (69, 5000, 50_000)
pe, inj, total = make_catalog(n_ev, n_pe, n_inj, seed=2025)
comp = COMPOSITIONS["plpeak"](pe, inj)
eng = comp.engine()
start = {"alpha": -2.5, "beta": 1.0, "mpp": 35.0, "sigpp": 5.0, "lam": 0.1, "lamb": 2.7}
theta0 = comp.theta(start)
names = [n for n, _ in comp._theta_map()]
idx = {n: i for i, n in enumerate(names)}
# priors of examples/simple_powerlaw_peak_example.py:52-77: Normal(0,5) on the slopes, Uniform(mmin,mmax)
# on the peak mean, HalfNormal(10) on its width, Uniform(0,1) on the mixing fraction
prior = GaussianSmoothingPrior(eng.n_theta)
for n in ("alpha", "beta", "lamb"):
    prior.sigmas[idx[n]] = 5.0
prior.sigmas[idx["sigpp"]] = 10.0
bij = Bijector(eng.n_theta).interval(idx["mpp"], 5.0, 100.0).interval(idx["lam"], 0.0, 1.0).positive(idx["sigpp"])
# min_neff_cut=False as in the reference's inference tests (tests/inference_test.py:185); `--neff-cut` applies the
# default cuts of analysis.py:272-303 (on a small catalog the posterior then hugs the cut and most trajectories
# end on it -- reported as divergences -- exactly as under numpyro)
target = make_target(eng, total, prior, bijector=bij, min_neff_cut="--neff-cut" in sys.argv)
if "--native" in sys.argv:
    lockstep = "--lockstep" in sys.argv
    per_group = min(16, n_chains)
    n_groups = (n_chains + per_group - 1) // per_group if lockstep else n_chains
    if lockstep:
        n_chains = n_groups * per_group
    engines = [eng] + [COMPOSITIONS["plpeak"](pe, inj).engine() for _ in range(n_groups - 1)]
    rng = np.random.default_rng(0)
    starts = np.stack([bij.forward(bij.inverse(theta0) + (0.05 * rng.normal(size=eng.n_theta) if c else 0.0))[0] for c in range(n_chains)])
    t0 = time.perf_counter()
    if lockstep:
        res = nuts_engine_lockstep(engines, per_group, total, prior, bij, starts, n_warmup=200, n_samples=200, seed=1, min_neff_cut="--neff-cut" in sys.argv)
    else:
        res = nuts_engine(engines, total, prior, bij, starts, n_warmup=200, n_samples=200, seed=1, min_neff_cut="--neff-cut" in sys.argv)
    dt = time.perf_counter() - t0
    n_ev_total = sum(r["n_evals"] for r in res)
    print(f"native sampler, {n_chains} chain(s){' in lock step' if lockstep else ''}: {n_ev_total} engine evaluations in {dt:.2f}s ({n_ev_total / dt:.0f} evals/s aggregate)")
    if lockstep:
        st = lockstep_stats()
        print(f"  {n_groups} group(s) of {per_group} chains ('{eng.batch_path(per_group)}' kernel): {st['batches']} batched evaluations of {st['mean_points_per_batch']:.1f} points on average")
        res = res[:8]  # (the per-chain lines below: the first eight)
    allth = np.concatenate([r["samples"] for r in res])
    for c, r in enumerate(res):
        print(f"  chain {c}: accept {r['accept_rate']:.2f}, step {r['step_size']:.3g}, mean tree depth {r['tree_depth'].mean():.1f}, {r['n_divergent']} divergent")
    for i, n in enumerate(names):
        print(f"  {n:8s} mean {allth[:, i].mean():9.3f}  sd {allth[:, i].std():8.3f}   chain means {np.round([r['samples'][:, i].mean() for r in res], 3)}")
    sys.exit(0)
if n_chains > 1:
    engines = [eng] + [COMPOSITIONS["plpeak"](pe, inj).engine() for _ in range(n_chains - 1)]
    pairs = [make_async_target(e, total, prior, bijector=bij, min_neff_cut="--neff-cut" in sys.argv) for e in engines]
    rng = np.random.default_rng(0)
    starts = [bij.inverse(theta0) + 0.05 * rng.normal(size=eng.n_theta) for _ in range(n_chains)]
    t0 = time.perf_counter()
    res = nuts_chains(pairs, starts, n_warmup=200, n_samples=200, seed=1)
    dt = time.perf_counter() - t0
    n_ev_total = sum(r["n_evals"] for r in res)
    print(f"{n_chains} chains: {n_ev_total} engine evaluations in {dt:.2f}s ({n_ev_total / dt:.0f} evals/s aggregate incl. the Python samplers)")
    allth = np.concatenate([np.array([bij.forward(u)[0] for u in r["samples"]]) for r in res])
    for c, r in enumerate(res):
        print(f"  chain {c}: accept {r['accept_rate']:.2f}, step {r['step_size']:.3g}, mean tree depth {r['tree_depth'].mean():.1f}, {r['n_divergent']} divergent")
    for i, n in enumerate(names):
        per_chain = [np.array([bij.forward(u)[0][i] for u in r["samples"]]).mean() for r in res]
        print(f"  {n:8s} mean {allth[:, i].mean():9.3f}  sd {allth[:, i].std():8.3f}   chain means {np.round(per_chain, 3)}")
    sys.exit(0)
t0 = time.perf_counter()
if use_hmc:
    out = hmc(target, bij.inverse(theta0), n_warmup=150, n_samples=150, n_leapfrog=8, seed=1, progress=50)
else:
    out = nuts(target, bij.inverse(theta0), n_warmup=200, n_samples=200, seed=1, progress=50)
out["samples"] = np.array([bij.forward(u)[0] for u in out["samples"]])
dt = time.perf_counter() - t0
print(f"{out['n_evals']} engine evaluations in {dt:.2f}s ({out['n_evals'] / dt:.0f} evals/s incl. the Python sampler), accept {out['accept_rate']:.2f}, step {out['step_size']:.3g}"
      + ("" if use_hmc else f", mean tree depth {out['tree_depth'].mean():.1f}, {out['n_divergent']} divergent"))
for i, n in enumerate(names):
    print(f"  {n:8s} mean {out['samples'][:, i].mean():9.3f}  sd {out['samples'][:, i].std():8.3f}")
