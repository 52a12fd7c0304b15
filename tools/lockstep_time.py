#!/usr/bin/env python3
"""Diagnostic (GPU box): the library's lock-step sampler on a bench configuration -- G groups of K chains.
  GWI_LOCKSTEP_STATS=1 python tools/lockstep_time.py c2 2 16 [n_warmup n_samples [n_chains]]
n_chains > G x K: a queue of chains over the G x K slots (gwi_nuts_engine_queue).  THREADED=T runs T chains at a time on T host
threads and engines instead (gwi_nuts_engine), the same chains in rounds, for comparison."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, reference_priors  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.sampling import lockstep_stats, nuts_engine, nuts_engine_lockstep  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

cfg, G, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
nw, ns = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (100, 50)
C = int(sys.argv[6]) if len(sys.argv) > 6 else G * K
comp_name, cat, _, _ = CONFIGS[cfg]
pe, inj, total = make_config_catalog(cat)
rng = np.random.default_rng(0)
comps = [COMPOSITIONS[comp_name](pe, inj) for _ in range(G)]
engs = [c.engine() for c in comps]
thetas = [comps[0].theta(draw_params(comp_name, rng)) for _ in range(C)]
prior, bij, _ = reference_priors(comp_name, comps[0], engs[0].n_theta)
starts = np.stack(thetas)
if os.environ.get("LOCKSTEP_SAME_START"):  # every chain from the same point (seeds differ)
    starts = np.stack([thetas[int(os.environ["LOCKSTEP_SAME_START"]) - 1]] * C)
if bij is not None:
    for k in np.flatnonzero(bij.kind == 3):
        starts[:, k] = bij.lo[k]
kw = dict(max_tree_depth=10, seed=1, min_neff_cut=False)
T = int(os.environ.get("THREADED", "0"))
if T:  # the same chains, T at a time on T threads (chain c keeps seed 1 + 1000 c: the same draws)
    while len(engs) < T:
        comps.append(COMPOSITIONS[comp_name](pe, inj))
        engs.append(comps[-1].engine())
    nuts_engine(engs[:T], total, prior, bij, starts[:T], n_warmup=2, n_samples=2, **dict(kw, max_tree_depth=4))
    t0 = time.perf_counter()
    res = []
    for c0 in range(0, C, T):
        n_now = min(T, C - c0)
        res += nuts_engine(engs[:n_now], total, prior, bij, starts[c0:c0 + n_now], n_warmup=nw, n_samples=ns, **dict(kw, seed=1 + 1000 * c0))
    dt = time.perf_counter() - t0
    how = f"{C} chains, {T} at a time on threads"
else:
    nuts_engine_lockstep(engs, K, total, prior, bij, starts[: G * K], n_warmup=2, n_samples=2, **dict(kw, max_tree_depth=4))
    t0 = time.perf_counter()
    res = nuts_engine_lockstep(engs, K, total, prior, bij, starts, n_warmup=nw, n_samples=ns, **kw)
    dt = time.perf_counter() - t0
    how = f"{C} chains over {G} x {K} slots, {lockstep_stats()['mean_points_per_batch']:.2f} points per batch"
n = sum(r["n_evals"] for r in res)
ev = np.array([r["n_evals"] for r in res])
print(f"{cfg} {how}: {n} evals in {dt:.2f} s = {n / dt:.0f} evals/s ({1e6 * dt / n:.2f} us per leapfrog); evals per chain min {ev.min()} max {ev.max()}; "
      f"divergences {sum(r['n_divergent'] for r in res)}, mean accept {np.mean([r['accept_rate'] for r in res]):.2f}", flush=True)
