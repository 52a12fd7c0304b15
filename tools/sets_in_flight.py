#!/usr/bin/env python3
"""Diagnostic (GPU box): T sets of K points in flight from ONE host thread (gwi_eval_batch_begin on set i + 1 before
gwi_eval_batch_end on set i, an engine per set), for engine variants given as environment overrides, interleaved.
  python tools/sets_in_flight.py c2 "" "GWI_PBATCH_BALANCED=0" """
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

cfg = sys.argv[1]
variants = [dict(kv.split("=") for kv in v.split()) if v.strip() else {} for v in sys.argv[2:]] or [{}]
K = int(os.environ.get("SIF_K", "16"))
comp_name, cat, _, _ = CONFIGS[cfg]
pe, inj, total = make_config_catalog(cat)
rng = np.random.default_rng(0)
runs = []
for env in variants:
    os.environ.update(env)
    comps = [COMPOSITIONS[comp_name](pe, inj) for _ in range(4)]
    engs = [c.engine() for c in comps]
    for k in env:
        os.environ.pop(k)
    runs.append((env, comps, engs))
tb = np.stack([runs[0][1][0].theta(draw_params(comp_name, rng)) for _ in range(K)])
for rep in range(3):
    for env, comps, engs in runs:
        row = []
        vg = engs[0].configure_batch(K, total, min_neff_cut=False)
        for _ in range(20):
            vg(tb)
        t0 = time.perf_counter()
        for _ in range(200):
            vg(tb)
        row.append(200 * K / (time.perf_counter() - t0))
        for T in (2, 3, 4):
            hv = [e.configure_batch_async(K, total, min_neff_cut=False) for e in engs[:T]]
            for lap in range(2):
                n_1 = 30 if lap == 0 else 200
                t0 = time.perf_counter()
                for j in range(T - 1):
                    hv[j][0](tb)
                for it in range(n_1 * T):
                    hv[(it + T - 1) % T][0](tb)
                    hv[it % T][1]()
                for j in range(T - 1):
                    hv[(n_1 * T + j) % T][1]()
                dt = time.perf_counter() - t0
            row.append((200 * T + T - 1) * K / dt)
        print(f"{cfg} K={K} {env} [{engs[0].batch_path(K)}]: blocking {row[0] / 1e3:.1f} k evals/s; sets in flight from one thread: 2: {row[1] / 1e3:.1f} k, 3: {row[2] / 1e3:.1f} k, 4: {row[3] / 1e3:.1f} k", flush=True)
