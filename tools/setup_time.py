#!/usr/bin/env python3
"""Diagnostic (GPU box): construction cost of a model + engine (the setup path) with the columns
computed on the DEVICE from the raw catalog (gwi_create_ingest: model objects record setup expressions, one HIP kernel
evaluates them) against the HOST path (NumPy evaluates the same expressions, gwi_create uploads the result).
  python tools/setup_time.py [c2 c3 c5 c5x10]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.engine import NativePopulationLikelihood  # noqa: E402
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
    n = pe["mass_1"].size + inj["mass_1"].size
    res = {}
    for rounds in range(2):  # second round: allocator pools, code objects and page tables are warm for both paths
        for path in ("host", "device"):
            t0 = time.perf_counter()
            comp = COMPOSITIONS[comp_name](pe, inj)
            p = comp.placeholder()
            wpe, winj, hv = comp.weights(p, True), comp.weights(p, False), comp.hypervolume(p)
            t1 = time.perf_counter()
            eng = NativePopulationLikelihood(wpe, winj, hv, device_setup=(path == "device"))
            t2 = time.perf_counter()
            res[path] = (t1 - t0, t2 - t1, t2 - t0)
            th = eng.bound.theta_of(comp.weights(draw_params(comp_name, np.random.default_rng(0)), True))
            r = eng.evaluate(th, total, min_neff_cut=False)
            res[path + "_ll"] = r.log_likelihood
            if rounds == 1 and path == "device":
                t4 = time.perf_counter()
                for _ in range(20):
                    eng.evaluate(th, total, min_neff_cut=False)
                t_eval = (time.perf_counter() - t4) / 20
            eng.close()
            del eng, comp, wpe, winj, hv
    h, d = res["host"], res["device"]
    print(f"{cfg}: {n} samples, composition {comp_name}")
    print(f"      host setup   : models {h[0]:.3f} s + bind / NumPy columns / upload / gwi_create {h[1]:.3f} s = {h[2]:.3f} s")
    print(f"      device setup : models {d[0]:.3f} s + raw upload / ingest kernel / gwi_create_ingest {d[1]:.3f} s = {d[2]:.3f} s   ({h[2] / d[2]:.2f}x)")
    print(f"      log-likelihood host-setup {res['host_ll']!r} device-setup {res['device_ll']!r} (rel. diff {abs(res['host_ll'] - res['device_ll']) / abs(res['host_ll']):.1e})")
    print(f"      one evaluation {1e3 * t_eval:.3f} ms: the device setup equals {d[2] / t_eval:.0f} evaluations")
