#!/usr/bin/env python3
"""Diagnostic (GPU box): what ONE rank of an 8-GPU run does per evaluation -- the engine of shard 0 of 8 (contiguous events + injection
slice) timed alone on one GPU: scan / tail kernel durations and the C-loop time per local evaluation (no exchange).
  python tools/shard_time.py [c2 c3 c5] [world]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

args = [a for a in sys.argv[1:] if not a.isdigit()] or ["c2", "c3", "c5"]
world = int(next((a for a in sys.argv[1:] if a.isdigit()), "8"))
for cfg in args:
    comp_name, cat, c_alg, _ = CONFIGS[cfg]
    pe, inj, total = make_config_catalog(cat)
    for w in (1, world):
        comp = COMPOSITIONS[comp_name](pe, inj)
        eng = comp.engine(rank=0, world=w)
        rng = np.random.default_rng(0)
        ths = np.stack([eng.bound.theta_of(comp.weights(draw_params(comp_name, rng), True)) for _ in range(32)])
        eng.evaluate_sequence(ths, total, min_neff_cut=False)
        _, _, kms = eng.evaluate_sequence(np.concatenate([ths] * 4), total, min_neff_cut=False, timing_every=2)
        sel = kms[:, 0] >= 0
        loop_us = 1e6 * eng.selftime(ths[0], total, n_iter=400, min_neff_cut=False)
        n = eng.n_ev * eng.n_pe + eng.n_inj
        print(f"{cfg} shard 0 of {w}: {eng.n_ev} events + {eng.n_inj} injections = {n} samples: scan {1e3 * np.median(kms[sel, 0]):.2f} us, tail {1e3 * np.median(kms[sel, 1] + np.maximum(kms[sel, 2], 0)):.2f} us, "
              f"C loop {loop_us:.2f} us per local evaluation ({eng.lib.gwi_scan_kernel_name(eng.handle).decode()})", flush=True)
        eng.close()
