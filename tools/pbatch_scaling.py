#!/usr/bin/env python3
"""Diagnostic (GPU box): where the time of a batched launch of a parametric model goes -- throughput or fixed costs.
Config 2's model on f x its catalog (events and injections scaled together), K points per launch: if the scan time is
a + b f, then b is what the chip needs per catalog at full occupancy (instruction issue) and a is start-up, load latency and the
tail in which the last workgroups run alone.   python tools/pbatch_scaling.py [K=16] [factors=1,2,3,4,6,8]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gwinferno_amd import _native as N  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import BASE_SEED, make_catalog  # noqa: E402

Ks = [int(k) for k in (sys.argv[1] if len(sys.argv) > 1 else "16").split(",")]
factors = [float(f) for f in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3,4,6,8").split(",")]
name = os.environ.get("PB_MODEL", "plpeak")
rows = []
for f in factors:
    n_ev, n_inj = int(round(69 * f)), int(round(50_000 * f))
    pe, inj, total = make_catalog(n_ev, 5000, n_inj, seed=BASE_SEED + 2)
    comp = COMPOSITIONS[name](pe, inj)
    eng = comp.engine()
    rng = np.random.default_rng(0)
    ths = np.ascontiguousarray(np.stack([comp.theta(draw_params(name, rng)) for _ in range(max(Ks))]))
    geo = eng.launch_geometry()
    for K in Ks:
        opt = eng._options(total, None, False, False, False)
        summ = (N.GwiSummary * K)()
        grads, lb, ln, lv, norms = np.zeros((K, eng.n_theta)), np.zeros((K, eng.n_ev)), np.zeros((K, eng.n_ev)), np.zeros((K, eng.n_ev)), np.zeros((K, 8))
        args = (eng.handle, N.as_dp(ths), K, C.byref(opt), summ, N.as_dp(grads), N.as_dp(lb), N.as_dp(ln), N.as_dp(lv), N.as_dp(norms))
        for _ in range(30):
            eng.lib.gwi_eval_batch(*args)
        eng.set_timing(1)
        ks = []
        for _ in range(60):
            eng.lib.gwi_eval_batch(*args)
            ks.append(eng.last_kernel_ms())
        eng.set_timing(0)
        ks = 1e3 * np.median(np.array(ks), axis=0)
        n = n_ev * 5000 + n_inj
        rows.append((K, f, n, ks[0]))
        print(f"{name} x{f:g} K={K:2d} [{eng.batch_path(K)}]: {n} samples, geometry {geo}: scan {ks[0]:7.2f} us  combine {ks[1]:6.2f}  final {ks[2]:6.2f}  -> {1e3 * ks[0] / (n * K / 1e3):.3f} ns per 1000 sample-points", flush=True)
    eng.close()
    del eng, comp, pe, inj
for K in Ks:
    sel = [(f, t) for k, f, n, t in rows if k == K]
    if len(sel) >= 2:
        A = np.array([[1.0, f] for f, _ in sel])
        a, b = np.linalg.lstsq(A, np.array([t for _, t in sel]), rcond=None)[0]
        print(f"K={K}: scan us = {a:.2f} + {b:.2f} x catalogs   (fixed part {a:.1f} us of the {sel[0][1]:.1f} us at x{sel[0][0]:g})")
