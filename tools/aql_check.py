#!/usr/bin/env python3
"""Diagnostic (GPU box): the AQL dispatch path against the HIP-stream path of the same library -- same bits, and the
time per evaluation of each (set GWI_AQL=0 in a second process for the HIP numbers).   python tools/aql_check.py c2"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
comp_name, cat, _, _ = CONFIGS[cfg]
pe, inj, total = make_config_catalog(cat)
comp = COMPOSITIONS[comp_name](pe, inj)
eng = comp.engine()
print(f"{cfg}: dispatch: {eng.dispatch_info()}", flush=True)
rng = np.random.default_rng(0)
ths = np.stack([comp.theta(draw_params(comp_name, rng)) for _ in range(8)])
r = [eng.evaluate(t, total, min_neff_cut=False) for t in ths]
print("log_l:", [f"{x.log_likelihood:.9f}" for x in r[:3]], flush=True)
eng.set_timing(2)  # mode 2: timed through the HIP stream -- same kernels, other queue
r2 = [eng.evaluate(t, total, min_neff_cut=False) for t in ths]
eng.set_timing(False)
same = all(a.log_likelihood == b.log_likelihood and np.array_equal(a.grad, b.grad) and np.array_equal(a.log_bfs, b.log_bfs) for a, b in zip(r, r2))
ms_hip = np.array(eng.last_kernel_ms())
eng.set_timing(1)
for t in ths:
    eng.evaluate(t, total, min_neff_cut=False)
ms_aql = np.array(eng.last_kernel_ms())
eng.set_timing(0)
print("AQL path == HIP-stream path, bit for bit:", same, "| last kernel us [scan, combine, final]: HIP events", np.round(1e3 * ms_hip, 2), " AQL dispatch timestamps", np.round(1e3 * ms_aql, 2), flush=True)
seq = np.concatenate([ths] * 250)
eng.evaluate_sequence(seq[:200], total, min_neff_cut=False)
t0 = time.perf_counter()
ll, g = eng.evaluate_sequence(seq, total, min_neff_cut=False)
dt = time.perf_counter() - t0
print(f"{cfg}: {len(seq) / dt:.0f} evals/s in the library's loop ({1e6 * dt / len(seq):.2f} us per evaluation); repeatable: {bool(np.all(ll[:8] == ll[8:16]))}", flush=True)
eng.close()
