#!/usr/bin/env python3
"""Diagnostic (GPU box): kernel durations [scan, combine, final] of a batched launch and the wall time per batch.
  python tools/batch_kernels.py c2 16"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

cfg, K = sys.argv[1], int(sys.argv[2])
comp_name, cat, _, _ = CONFIGS[cfg]
pe, inj, total = make_config_catalog(cat)
comp = COMPOSITIONS[comp_name](pe, inj)
eng = comp.engine()
rng = np.random.default_rng(0)
tb = np.stack([comp.theta(draw_params(comp_name, rng)) for _ in range(K)])
vgb = eng.configure_batch(K, total, min_neff_cut=False)
for _ in range(20):
    vgb(tb)
t0 = time.perf_counter()
for _ in range(200):
    vgb(tb)
wall = (time.perf_counter() - t0) / 200
eng.set_timing(1)
ms = []
for _ in range(20):
    vgb(tb)
    ms.append(eng.last_kernel_ms())
eng.set_timing(0)
ms = np.mean(np.array(ms), axis=0)
print(f"{cfg} K={K} path={eng.batch_path(K)} dispatch={eng.dispatch_info()}: wall {1e6 * wall:.1f} us/batch ({1e6 * wall / K:.2f} us/eval); kernels scan {1e3 * ms[0]:.1f} combine {1e3 * ms[1]:.1f} final {1e3 * ms[2]:.1f} us")
