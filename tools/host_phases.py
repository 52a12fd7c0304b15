#!/usr/bin/env python3
"""Diagnostic (GPU box): where the HOST spends one sequential evaluation (build with -DGWI_HOST_PHASES):
  hipcc ... -DGWI_HOST_PHASES ... -o gwinferno_amd/_lib/exp/lib_phases.so
  GWI_ENGINE_LIB=gwinferno_amd/_lib/exp/lib_phases.so python tools/host_phases.py c2"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

for cfg in sys.argv[1:]:
    comp_name, cat, _, _ = CONFIGS[cfg]
    pe, inj, total = make_config_catalog(cat)
    comp = COMPOSITIONS[comp_name](pe, inj)
    eng = comp.engine()
    th = comp.theta(draw_params(comp_name, np.random.default_rng(0)))
    eng.selftime(th, total, n_iter=300, min_neff_cut=False)
    out = (C.c_double * 6)()
    eng.lib.gwi_debug_host_phases(out)
    loop = 1e6 * eng.selftime(th, total, n_iter=2000, min_neff_cut=False)
    eng.lib.gwi_debug_host_phases(out)
    n = out[5]
    names = ["prelude", "scan launch", "combine launch", "norm launch", "wait + host sums"]
    print(f"{cfg}: C loop {loop:.2f} us/eval; host phases (us): " + ", ".join(f"{nm} {1e6 * out[i] / n:.2f}" for i, nm in enumerate(names)), flush=True)
    eng.close()
