#!/usr/bin/env python3
"""Diagnostic (GPU box): the library's lock-step sampler on a bench configuration -- G groups of K chains.
  GWI_LOCKSTEP_STATS=1 python tools/lockstep_time.py c2 2 16 [n_warmup n_samples]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, reference_priors  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.sampling import nuts_engine_lockstep  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

cfg, G, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
nw, ns = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (100, 50)
comp_name, cat, _, _ = CONFIGS[cfg]
pe, inj, total = make_config_catalog(cat)
rng = np.random.default_rng(0)
comps = [COMPOSITIONS[comp_name](pe, inj) for _ in range(G)]
engs = [c.engine() for c in comps]
thetas = [comps[0].theta(draw_params(comp_name, rng)) for _ in range(G * K)]
prior, bij, _ = reference_priors(comp_name, comps[0], engs[0].n_theta)
starts = np.stack(thetas)
if os.environ.get("LOCKSTEP_SAME_START"):  # every chain from the same point (seeds differ)
    starts = np.stack([thetas[int(os.environ["LOCKSTEP_SAME_START"]) - 1]] * (G * K))
if bij is not None:
    for k in np.flatnonzero(bij.kind == 3):
        starts[:, k] = bij.lo[k]
kw = dict(max_tree_depth=10, seed=1, min_neff_cut=False)
nuts_engine_lockstep(engs, K, total, prior, bij, starts, n_warmup=2, n_samples=2, **dict(kw, max_tree_depth=4))
t0 = time.perf_counter()
res = nuts_engine_lockstep(engs, K, total, prior, bij, starts, n_warmup=nw, n_samples=ns, **kw)
dt = time.perf_counter() - t0
n = sum(r["n_evals"] for r in res)
ev = np.array([r["n_evals"] for r in res])
print(f"{cfg} {G} x {K} chains: {n} evals in {dt:.2f} s = {n / dt:.0f} evals/s ({1e6 * dt / n:.2f} us per leapfrog); evals per chain min {ev.min()} max {ev.max()}", flush=True)
