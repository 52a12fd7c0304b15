#!/usr/bin/env python3
"""Diagnostic (GPU box): throughput of batched evaluation sets (K points per gwi_eval_batch) when T host threads each drive an
engine of their own on the same catalog -- how much of a blocking batch's tail (combine, final, host) other sets' scans can hide.
  python tools/batch_overlap.py c2 16 1 2 3"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd import _native as N  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

cfg, K = sys.argv[1], int(sys.argv[2])
counts = [int(x) for x in sys.argv[3:]] or [1, 2]
comp_name, cat, _, _ = CONFIGS[cfg]
pe, inj, total = make_config_catalog(cat)
rng = np.random.default_rng(0)
comps = [COMPOSITIONS[comp_name](pe, inj) for _ in range(max(counts))]
engs = [c.engine() for c in comps]
ths = np.ascontiguousarray(np.stack([comps[0].theta(draw_params(comp_name, rng)) for _ in range(K)]))


def make_call(eng):
    opt = eng._options(total, None, False, False, False)
    summ = (N.GwiSummary * K)()
    grads = np.zeros((K, eng.n_theta))
    args = (eng.handle, N.as_dp(ths), K, C.byref(opt), summ, N.as_dp(grads), None, None, None, None)
    return lambda: eng.lib.gwi_eval_batch(*args), (opt, summ, grads)


calls = [make_call(e) for e in engs]
for T in counts:
    n = 400
    gate = threading.Barrier(T + 1)

    def work(i):
        f = calls[i][0]
        for _ in range(30):
            f()
        gate.wait()
        for _ in range(n):
            f()
        gate.wait()

    ws = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    for w in ws:
        w.start()
    gate.wait()
    t0 = time.perf_counter()
    gate.wait()
    dt = time.perf_counter() - t0
    for w in ws:
        w.join()
    print(f"{cfg} K={K} sets in flight {T}: {T * n * K / dt:9.0f} evals/s  ({1e6 * dt / (T * n * K):.2f} us/eval)", flush=True)
