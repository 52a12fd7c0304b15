#!/usr/bin/env python3
"""Diagnostic (GPU box): cost of ONE call of a model function written against the drop-in API (model objects -> lazy
products -> hierarchical_likelihood) without JAX, next to the engine's own time per evaluation."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gwinferno_amd import likelihood as L  # noqa: E402
from gwinferno_amd.lazy import where_finite  # noqa: E402
from gwinferno_amd.models import PowerlawRedshiftModel, plpeak_primary_ratio_pdf  # noqa: E402
from gwinferno_amd.synthetic import make_config_catalog  # noqa: E402

pe, inj, total = make_config_catalog("c2")
z_model = PowerlawRedshiftModel(z_pe=pe["redshift"], z_inj=inj["redshift"])
L.SAMPLE_VALUES["unscaled_rate"] = 30.0


def model(alpha, beta, mpp, sigpp, lam, lamb):  # tests/inference_test.py:162-197 with the PL+Peak mass model
    def w(d):
        return where_finite(plpeak_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], alpha, beta, 5.0, 100.0, mpp, sigpp, lam) * z_model(d["redshift"], lamb) / d["prior"])

    return L.hierarchical_likelihood(w(pe), w(inj), total_inj=total, Nobs=pe["mass_1"].shape[0], Tobs=1.0, surveyed_hypervolume=z_model.normalization(lamb), min_neff_cut=False)


rng = np.random.default_rng(0)
pts = [dict(alpha=rng.normal(-2.5, 0.3), beta=rng.normal(1, 0.3), mpp=rng.uniform(25, 45), sigpp=rng.uniform(2, 8), lam=rng.uniform(0.02, 0.2), lamb=rng.normal(2.7, 0.5)) for _ in range(64)]
for p in pts[:20]:
    model(**p)
n = 3000
t0 = time.perf_counter()
for i in range(n):
    model(**pts[i & 63])
dt = time.perf_counter() - t0
print(f"drop-in model function: {1e6 * dt / n:.1f} us per call ({n / dt:.0f} calls/s); log_likelihood {L.last_sites()['log_likelihood']:.6f}")
