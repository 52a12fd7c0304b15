#!/usr/bin/env python3
"""Diagnostic (GPU box): C host threads, one engine each, every thread a blocking evaluation loop (ctypes
releases the GIL inside gwi_eval): aggregate evaluations/s of independent chains when the launch cost is
spread over cores.   python tools/threads.py c2 1 2 4 8"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

cfg = sys.argv[1]
counts = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
comp_name, cat, _, _ = CONFIGS[cfg]
pe, inj, total = make_config_catalog(cat)
rng = np.random.default_rng(0)
comps = [COMPOSITIONS[comp_name](pe, inj) for _ in range(max(counts))]
engines = [c.engine() for c in comps]
thetas = [comps[0].theta(draw_params(comp_name, rng)) for _ in range(64)]
for C in counts:
    n = 4000
    go = threading.Barrier(C + 1)

    def work(eng, off):
        th = thetas[off % 64]
        eng.selftime(th, total, n_iter=50, min_neff_cut=False)
        go.wait()
        eng.selftime(th, total, n_iter=n, min_neff_cut=False)  # C loop of n blocking evaluations, GIL released
        go.wait()

    ts = [threading.Thread(target=work, args=(engines[c], c)) for c in range(C)]
    for t in ts:
        t.start()
    go.wait()
    t0 = time.perf_counter()
    go.wait()
    dt = time.perf_counter() - t0
    for t in ts:
        t.join()
    print(f"{cfg}: {C} thread(s): {C * n / dt:9.0f} evals/s aggregate ({1e6 * dt / (C * n):.2f} us per evaluation)", flush=True)
