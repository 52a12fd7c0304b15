#!/usr/bin/env python3
"""Diagnostic (GPU box): wall time of the posterior-predictive mass curves for N posterior draws
(800 x 800 mesh per draw, both marginals), engine construction included."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gwinferno_amd import postprocess as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(0)
t0 = time.perf_counter()
P.calculate_powerlaw_peak_mass_ppds(rng.normal(-2.5, 0.7, n), rng.normal(1.0, 0.7, n), rng.uniform(25, 45, n), rng.uniform(2, 8, n), rng.uniform(0.02, 0.2, n), 5.0, 100.0)
t1 = time.perf_counter()
P.calculate_bspline_mass_ppds(rng.normal(size=(n, 30)), rng.normal(size=(n, 14)), {"m1": 30, "q": 14}, 5.0, 100.0)
t2 = time.perf_counter()
print(f"{n} draws: power-law+peak mass PPDs {t1 - t0:.2f} s ({1e3 * (t1 - t0) / n:.2f} ms/draw); B-spline(30, 14) mass PPDs {t2 - t1:.2f} s ({1e3 * (t2 - t1) / n:.2f} ms/draw)")
