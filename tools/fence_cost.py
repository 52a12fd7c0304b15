#!/usr/bin/env python3
"""Diagnostic (GPU box): what the bench's bracket costs a 20-step timed region -- torch.cuda.synchronize() after the engine's
own stream / AQL queue have been busy, and the fixed cost of one gwi_eval_sequence call."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

comp_name, cat, _, _ = CONFIGS["c2"]
pe, inj, total = make_config_catalog(cat)
comp = COMPOSITIONS[comp_name](pe, inj)
eng = comp.engine()
rng = np.random.default_rng(0)
ths = np.stack([comp.theta(draw_params(comp_name, rng)) for _ in range(64)])
eng.evaluate_sequence(np.concatenate([ths] * 200), total, min_neff_cut=False)  # clocks up
for K in (1, 20, 200, 2000):
    seq = np.stack([ths[i % 64] for i in range(K)])
    timed = eng.configure_sequence(seq, total, min_neff_cut=False)
    rows = []
    for rep in range(30):
        eng.evaluate_sequence(seq[: min(K, 5)], total, min_neff_cut=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        timed()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1))
    r = np.array(rows)
    print(f"K={K:5d}: call {1e6 * np.median(r[:, 0]):9.1f} us ({1e6 * np.median(r[:, 0]) / K:6.2f} per step), closing synchronize {1e6 * np.median(r[:, 1]):6.1f} us (p90 {1e6 * np.percentile(r[:, 1], 90):6.1f})"
          f" -> bracketed {1e6 * np.median(r.sum(1)) / K:6.2f} us per step", flush=True)
