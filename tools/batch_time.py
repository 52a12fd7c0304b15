#!/usr/bin/env python3
"""Diagnostic (GPU box): time of one gwi_eval_batch call (K points, preallocated buffers, no Python result
objects) for engine variants given as environment overrides, interleaved in one process.
  python tools/batch_time.py c2 "GWI_STAGE_KERNEL=0" "GWI_STAGE_KERNEL=1" """
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd import _native as N  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

cfg = sys.argv[1]
variants = [dict(kv.split("=") for kv in v.split()) if v.strip() else {} for v in sys.argv[2:]] or [{}]
comp_name, cat, _, _ = CONFIGS[cfg]
pe, inj, total = make_config_catalog(cat)
rng = np.random.default_rng(0)
runs = []
for env in variants:
    os.environ.update(env)
    comp = COMPOSITIONS[comp_name](pe, inj)
    eng = comp.engine()
    for k in env:
        os.environ.pop(k)
    runs.append((env, comp, eng))
ths = np.ascontiguousarray(np.stack([runs[0][1].theta(draw_params(comp_name, rng)) for _ in range(64)]))
for K in [int(k) for k in os.environ.get("BT_KS", "1,4,16").split(",")]:
    for rep in range(2):
        for env, comp, eng in runs:
            opt = eng._options(total, None, False, False, False)
            summ = (N.GwiSummary * K)()
            grads, lb, ln, lv, norms = np.zeros((K, eng.n_theta)), np.zeros((K, eng.n_ev)), np.zeros((K, eng.n_ev)), np.zeros((K, eng.n_ev)), np.zeros((K, 8))
            args = (eng.handle, N.as_dp(ths), K, C.byref(opt), summ, N.as_dp(grads), N.as_dp(lb), N.as_dp(ln), N.as_dp(lv), N.as_dp(norms))
            for _ in range(30):
                eng.lib.gwi_eval_batch(*args)
            n = 400
            t0 = time.perf_counter()
            for _ in range(n):
                eng.lib.gwi_eval_batch(*args)
            dt = time.perf_counter() - t0
            eng.set_timing(1)
            ks = []
            for _ in range(40):
                eng.lib.gwi_eval_batch(*args)
                ks.append(eng.last_kernel_ms())
            eng.set_timing(0)
            ks = 1e3 * np.median(np.array(ks), axis=0)
            print(f"{cfg} K={K:2d} {env} [{eng.batch_path(K)}]: {1e6 * dt / n:7.1f} us/batch  {1e6 * dt / n / K:6.2f} us/eval   scan/combine/final us {np.round(ks, 1)}", flush=True)
