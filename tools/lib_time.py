#!/usr/bin/env python3
"""Diagnostic (GPU box): time ONE build of the engine library on one or more configs.
  GWI_ENGINE_LIB=gwinferno_amd/_lib/exp/lib_x.so python tools/lib_time.py c3 c5
Prints the C-loop time per evaluation and the HIP-event durations [scan, combine, final] in us.
Compare builds only from runs inside the same gpurun call (box-to-box variation is ~10 %)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

tag = os.path.basename(os.environ.get("GWI_ENGINE_LIB", "default"))
for cfg in sys.argv[1:]:
    comp_name, cat, _, _ = CONFIGS[cfg]
    pe, inj, total = make_config_catalog(cat)
    comp = COMPOSITIONS[comp_name](pe, inj)
    eng = comp.engine()
    th = comp.theta(draw_params(comp_name, np.random.default_rng(0)))
    loops = []
    for _ in range(3):
        eng.selftime(th, total, n_iter=100, min_neff_cut=False)
        loops.append(1e6 * eng.selftime(th, total, n_iter=int(os.environ.get("AB_ITERS", "500")), min_neff_cut=False))
    eng.set_timing(True)
    ks = []
    for _ in range(200):
        eng.evaluate(th, total, min_neff_cut=False)
        ks.append(eng.last_kernel_ms())
    ks = 1e3 * np.median(np.array(ks), axis=0)
    r = eng.evaluate(th, total, min_neff_cut=False)
    print(f"{tag:24s} {cfg}: loop us/eval {np.median(loops):8.2f}  scan/combine/final us {np.round(ks, 2)}  log_l {r.log_likelihood:.6f}", flush=True)
    eng.close()
