#!/usr/bin/env python3
"""Diagnostic (GPU box): construction cost of a model + engine (the setup path) --
model objects (masks, coordinates, grids), bind (columns, kappa), gwi_create (upload)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS  # noqa: E402
from gwinferno_amd.engine import NativePopulationLikelihood, bind  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

import torch  # noqa: E402,F401  (first import and HIP start-up are not what is being timed)

torch.cuda.init()
for cfg in sys.argv[1:] or ["c2", "c3", "c5"]:
    if cfg == "c5x10":  # ten times BASELINE config 5: 2000 events x 10 000 PE + 5 M injections (25 M samples)
        from gwinferno_amd.synthetic import BASE_SEED, make_catalog

        comp_name = "bspline_full"
        pe, inj, total = make_catalog(2000, 10_000, 5_000_000, seed=BASE_SEED + 50)
    else:
        comp_name, cat, _, _ = CONFIGS[cfg]
        pe, inj, total = make_config_catalog(cat)
    t0 = time.perf_counter()
    comp = COMPOSITIONS[comp_name](pe, inj)
    p = comp.placeholder()
    wpe, winj, hv = comp.weights(p, True), comp.weights(p, False), comp.hypervolume(p)
    t1 = time.perf_counter()
    bm = bind(wpe, winj, hv)
    t2 = time.perf_counter()
    eng = NativePopulationLikelihood(wpe, winj, hv)
    t3 = time.perf_counter()
    n = pe["mass_1"].size + inj["mass_1"].size
    print(f"{cfg}: {n} samples, {len(bm.pe_cols)} columns: models {t1 - t0:.3f} s, bind {t2 - t1:.3f} s, bind + gwi_create {t3 - t2:.3f} s  (total {t3 - t0:.3f} s)")
    import numpy as np

    from gwinferno_amd.compositions import draw_params

    th = comp.theta(draw_params(comp_name, np.random.default_rng(0)))
    eng.evaluate(th, total, min_neff_cut=False)
    t4 = time.perf_counter()
    for _ in range(20):
        eng.evaluate(th, total, min_neff_cut=False)
    t_eval = (time.perf_counter() - t4) / 20
    print(f"      one evaluation {1e3 * t_eval:.3f} ms: the setup equals {(t3 - t0) / t_eval:.0f} evaluations (a 200 + 200 iteration NUTS run makes ~10^4-10^5)")
    eng.close()
