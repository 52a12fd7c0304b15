#!/usr/bin/env python3
"""Diagnostic (GPU box): scan / tail / C-loop time over a spread of catalog shapes, to spot launch geometries that fall off a cliff.
  python tools/shape_survey.py [plpeak bspline_iid bspline_full]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_catalog  # noqa: E402

SHAPES = [(10, 1000, 5000), (3, 100_000, 10_000), (1000, 500, 100_000), (69, 20_000, 1_000_000), (2000, 100, 50_000), (1, 5000, 5000), (400, 5000, 200_000), (25, 40_000, 100_000)]
for name in sys.argv[1:] or ["plpeak", "bspline_iid"]:
    for n_ev, n_pe, n_inj in SHAPES:
        pe, inj, total = make_catalog(n_ev, n_pe, n_inj, seed=11)
        comp = COMPOSITIONS[name](pe, inj)
        eng = comp.engine()
        rng = np.random.default_rng(0)
        ths = np.stack([eng.bound.theta_of(comp.weights(draw_params(name, rng), True)) for _ in range(16)])
        eng.evaluate_sequence(ths, total, min_neff_cut=False)
        _, _, kms = eng.evaluate_sequence(np.concatenate([ths] * 4), total, min_neff_cut=False, timing_every=2)
        sel = kms[:, 0] >= 0
        loop_us = 1e6 * eng.selftime(ths[0], total, n_iter=200, min_neff_cut=False)
        n = n_ev * n_pe + n_inj
        g = eng.launch_geometry()
        scan = 1e3 * np.median(kms[sel, 0])
        print(f"{name:12s} {n_ev:5d} ev x {n_pe:6d} PE + {n_inj:7d} inj = {n:8d}: scan {scan:7.2f} us ({1e3 * scan / n:6.2f} ns/k-sample), tail {1e3 * np.median(kms[sel, 1] + np.maximum(kms[sel, 2], 0)):6.2f}, "
              f"C loop {loop_us:7.2f} us | tiles/event {g['tiles_per_event']:3d} of {g['chunk_pe']:5d}, inj tiles {g['n_inj_tiles']:4d} of {g['chunk_inj']:5d}, workgroups {g['n_scan_blocks']:5d}", flush=True)
        eng.close()
        del eng, comp, pe, inj
