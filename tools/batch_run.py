#!/usr/bin/env python3
"""A bare loop of batched evaluations (gwi_eval_batch, K points per launch) for profiling the batched scan kernels:
    python3 tools/batch_run.py --config c5 --k 16 --n 40 [--mfma]
--mfma selects the matrix-core path (GWI_BATCH_MFMA=1: gwinferno_amd/csrc/gwi_mfma.h).  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c5")
ap.add_argument("--k", type=int, default=16)
ap.add_argument("--n", type=int, default=40)
ap.add_argument("--mfma", action="store_true")
ap.add_argument("--rows", action="store_true")
args = ap.parse_args()
if args.mfma:
    os.environ["GWI_BATCH_MFMA"] = "1"
if args.rows:
    os.environ["GWI_BATCH_ROWS"] = "1"
from gwinferno_amd.compositions import COMPOSITIONS, draw_params  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

from bench import CONFIGS  # noqa: E402

name, cat = CONFIGS[args.config][:2]
pe, inj, total = make_config_catalog(cat)
comp = COMPOSITIONS[name](pe, inj)
eng = comp.engine()
rng = np.random.default_rng(1234)
tb = np.stack([comp.theta(draw_params(name, rng)) for _ in range(args.k)])
vgb = eng.configure_batch(args.k, total, min_neff_cut=False)
for _ in range(5):
    vgb(tb)
t0 = time.perf_counter()
for _ in range(args.n):
    vgb(tb)
dt = time.perf_counter() - t0
print(json.dumps({"config": args.config, "k_batch": args.k, "path": eng.batch_path(args.k), "us_per_eval": 1e6 * dt / (args.n * args.k), "evals_per_s": args.n * args.k / dt,
                  "samples": int(eng.n_ev * eng.n_pe + eng.n_inj), "n_theta": int(eng.n_theta)}))
eng.close()
