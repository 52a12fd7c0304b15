#!/usr/bin/env python3
"""Diagnostic (GPU box): scan time of catalogs beyond the 256 MB Infinity Cache -- a BASELINE configuration's model on a multiple
of its catalog (events and injections scaled together).  python tools/big_catalog_time.py c5:2 c5:10 c2:25 c3:25"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import BASE_SEED, CONFIG_SIZES, make_catalog  # noqa: E402

for arg in sys.argv[1:] or ["c5:2", "c5:10"]:
    cfg, f = arg.split(":")
    f = int(f)
    comp_name, cat, c_alg, _ = CONFIGS[cfg]
    _, n_ev, n_pe, n_inj = CONFIG_SIZES[cat]
    pe, inj, total = make_catalog(n_ev * f, n_pe, n_inj * f, seed=BASE_SEED + 50)
    comp = COMPOSITIONS[comp_name](pe, inj)
    eng = comp.engine()
    rng = np.random.default_rng(0)
    ths = np.stack([comp.theta(draw_params(comp_name, rng)) for _ in range(16)])
    eng.evaluate_sequence(ths, total, min_neff_cut=False)
    _, _, kms = eng.evaluate_sequence(np.concatenate([ths] * 2), total, min_neff_cut=False, timing_every=1)
    sel = kms[:, 0] >= 0
    scan_us = 1e3 * float(np.median(kms[sel, 0]))
    n = pe["mass_1"].size + inj["mass_1"].size
    b = 8 * c_alg * n
    print(f"{cfg} x{f}: {n} samples, {b / 1e6:.0f} MB algorithmic ({8 * c_alg} B per sample): scan {scan_us:.1f} us = {b / scan_us / 1e6:.2f} TB/s ({b / scan_us / 8e6:.3f} of 8 TB/s); "
          f"tail {1e3 * float(np.median(kms[sel, 1] + np.maximum(kms[sel, 2], 0))):.1f} us; kernel {eng.lib.gwi_scan_kernel_name(eng.handle).decode()}", flush=True)
    eng.close()
    del eng, comp, pe, inj
