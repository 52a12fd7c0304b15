#!/usr/bin/env python3
"""A/B timing of engine variants on ONE GPU in ONE process (run-to-run and box-to-box variation is
~10 %, so variants must be interleaved).  Each variant is a dict of environment overrides applied while
its engine is created.   python tools/ab_bench.py c2 "GWI_SAMPLES_PER_LANE=2" "GWI_SAMPLES_PER_LANE=1 GWI_SINGLE_ROUND=1" """
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

cfg = sys.argv[1]
variants = [dict(kv.split("=") for kv in v.split()) if v.strip() else {} for v in sys.argv[2:]] or [{}]
comp_name, cat, _, _ = CONFIGS[cfg]
pe, inj, total = make_config_catalog(cat)
engines = []
for env in variants:
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    comp = COMPOSITIONS[comp_name](pe, inj)
    eng = comp.engine()
    engines.append((comp, eng))
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
th = engines[0][0].theta(draw_params(comp_name, np.random.default_rng(0)))
n_iter = int(os.environ.get("AB_ITERS", "1500"))
res = [[] for _ in engines]
for rep in range(int(os.environ.get("AB_REPS", "5"))):
    for i, (comp, eng) in enumerate(engines):
        eng.selftime(th, total, n_iter=200, min_neff_cut=False)
        res[i].append(1e6 * eng.selftime(th, total, n_iter=n_iter, min_neff_cut=False))
for env, r in zip(variants, res):
    print(f"{cfg} {env}: C-loop us/eval  median {np.median(r):7.2f}  min {min(r):7.2f}  all {np.round(r, 2)}")
