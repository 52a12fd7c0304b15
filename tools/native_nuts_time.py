#!/usr/bin/env python3
"""Diagnostic (GPU box): likelihood evaluations per second INSIDE a sampler -- the library's C++ NUTS
(gwi_nuts_engine, one host thread and one engine per chain) against the NumPy NUTS of gwinferno_amd.sampling on
the same target.   python tools/native_nuts_time.py c2 1 2 4"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.sampling import GaussianSmoothingPrior, make_target, nuts, nuts_engine  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

cfg = sys.argv[1]
counts = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
comp_name, cat, _, _ = CONFIGS[cfg]
pe, inj, total = make_config_catalog(cat)
rng = np.random.default_rng(0)
comps = [COMPOSITIONS[comp_name](pe, inj) for _ in range(max(counts))]
engines = [c.engine() for c in comps]
n = engines[0].n_theta
theta0 = comps[0].theta(draw_params(comp_name, rng))
# spline models: the reference's coefficient prior scale N(0, 1); a 10 times wider one measures the two-pass repeat path (bench.py:multi_chain)
prior = GaussianSmoothingPrior(n).normal(slice(0, n), 1.0 if comp_name.startswith("bspline") else 10.0)
kw = dict(n_warmup=150, n_samples=150, max_tree_depth=6, seed=1)
flags = dict(min_neff_cut=False)

t0 = time.perf_counter()
out = nuts(make_target(engines[0], total, prior, **flags), theta0, **kw)
dt = time.perf_counter() - t0
print(f"{cfg}: numpy NUTS, 1 chain : {out['n_evals'] / dt:9.0f} evals/s ({1e6 * dt / out['n_evals']:.1f} us per leapfrog), accept {out['accept_rate']:.2f}", flush=True)
for C in counts:
    starts = np.stack([theta0 * (1 + 0.01 * c) for c in range(C)])
    nuts_engine(engines[:C], total, prior, None, starts, **flags, **dict(kw, n_warmup=5, n_samples=5))
    t0 = time.perf_counter()
    res = nuts_engine(engines[:C], total, prior, None, starts, **flags, **kw)
    dt = time.perf_counter() - t0
    ev = sum(r["n_evals"] for r in res)
    print(f"{cfg}: native NUTS, {C} chain(s): {ev / dt:9.0f} evals/s aggregate ({1e6 * dt / ev:.2f} us per leapfrog), accept "
          + " ".join(f"{r['accept_rate']:.2f}" for r in res) + f", divergent {sum(r['n_divergent'] for r in res)}", flush=True)
