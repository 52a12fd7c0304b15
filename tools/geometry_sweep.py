#!/usr/bin/env python3
"""Diagnostic (GPU box): scan-kernel duration and C-loop time per evaluation for a list of launch geometries of one
BASELINE config, interleaved in one process (box-to-box spread is ~10 %).  Each variant is a string of environment
overrides read at gwi_create (GWI_PE_CHUNK / GWI_INJ_CHUNK / GWI_SAMPLES_PER_LANE / ...).
  python tools/geometry_sweep.py c5 "" "GWI_PE_CHUNK=2500 GWI_INJ_CHUNK=2440" """
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS  # noqa: E402
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

cfg = sys.argv[1]
variants = [dict(kv.split("=") for kv in v.split()) if v.strip() else {} for v in sys.argv[2:]] or [{}]
comp_name, cat, _, _ = CONFIGS[cfg]
pe, inj, total = make_config_catalog(cat)
rng = np.random.default_rng(0)
engines = []
for env in variants:
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    comp = COMPOSITIONS[comp_name](pe, inj)
    try:
        eng = comp.engine()
    except Exception as exc:  # a geometry the engine refuses
        print(f"{cfg} {env}: {exc}", flush=True)
        eng = None
    engines.append((comp, eng))
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
ths = np.stack([engines[0][0].theta(draw_params(comp_name, rng)) for _ in range(64)])
n_iter = int(os.environ.get("SWEEP_ITERS", "400"))
scan = [[] for _ in engines]
comb = [[] for _ in engines]
loop = [[] for _ in engines]
for rep in range(int(os.environ.get("SWEEP_REPS", "3"))):
    for i, (comp, eng) in enumerate(engines):
        if eng is None:
            continue
        eng.evaluate_sequence(ths, total, min_neff_cut=False)
        _, _, kms = eng.evaluate_sequence(np.concatenate([ths] * 4), total, min_neff_cut=False, timing_every=4)
        sel = kms[:, 0] >= 0
        scan[i].append(1e3 * float(np.mean(kms[sel, 0])))
        comb[i].append(1e3 * float(np.mean(kms[sel, 1] + np.maximum(kms[sel, 2], 0))))
        loop[i].append(1e6 * eng.selftime(ths[0], total, n_iter=n_iter, min_neff_cut=False))
for env, s, c, l, (comp, eng) in zip(variants, scan, comb, loop, engines):
    if eng is None:
        continue
    print(f"{cfg} {env}: scan us median {np.median(s):7.2f} min {min(s):7.2f} | tail us {np.median(c):6.2f} | C-loop us/eval median {np.median(l):7.2f} min {min(l):7.2f} | "
          f"repeats {eng.two_pass_repeats()}", flush=True)
