#!/usr/bin/env python3
"""Diagnostic (GPU box): soak test of the completion-stamp protocol -- many sequential evaluations, single and
batched, checking every result against the first (same theta => same bits for values)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
comp_name, cat, _, _ = CONFIGS[cfg]
pe, inj, total = make_config_catalog(cat)
comp = COMPOSITIONS[comp_name](pe, inj)
eng = comp.engine()
rng = np.random.default_rng(0)
ths = np.stack([comp.theta(draw_params(comp_name, rng)) for _ in range(16)])
vg = eng.configure(total, min_neff_cut=False)
ref = [vg(t)[0] for t in ths]
t0 = time.perf_counter()
bad = 0
for i in range(n):
    v, _ = vg(ths[i & 15])
    bad += v != ref[i & 15]
dt = time.perf_counter() - t0
print(f"{cfg}: {n} sequential evaluations in {dt:.1f} s ({n / dt:.0f}/s), value mismatches: {bad}")
vgb = eng.configure_batch(16, total, min_neff_cut=False)
refb = vgb(ths)[0].copy()
t0 = time.perf_counter()
badb = 0
for i in range(n // 32):
    v, _ = vgb(ths)
    badb += int(np.sum(v != refb))
dt = time.perf_counter() - t0
print(f"{cfg}: {n // 32} batches of 16 in {dt:.1f} s ({(n // 32) * 16 / dt:.0f} evals/s), value mismatches: {badb}; single vs batched max rel diff {np.max(np.abs(refb - np.array(ref)) / np.abs(np.array(ref))):.2e}")
