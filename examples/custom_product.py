#!/usr/bin/env python3
"""A product of densities nobody compiled ahead of time.

GWInferno is a toolkit: a model function multiplies whatever population models the user picks
(tests/inference_test.py:256-260, gwinferno/models/bsplines/separable.py:295-778).  This one -- PL+Peak primary mass x power-law
mass ratio (parametric.py:39-46), independent Beta spin magnitudes (:71-81), IID B-spline spin tilts (separable.py:156-218),
power-law redshift (parametric.py:112-145) -- has the sorted term-kind sequence 2,3,4,4,6,7,7, for which the library ships no
scan kernel.  `gwi_create` has hipRTC instantiate the scan template for exactly that sequence (1-2 s, once per machine: the code
object is kept under ~/.cache/gwinferno_amd), and the engine runs it like any ahead-of-time chain: through its AQL queue,
batched (the library's samplers take it like any other engine: examples/sample_plpeak_hmc.py).

    python examples/custom_product.py [n_events n_pe n_inj]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gwinferno_amd import models as M  # noqa: E402
from gwinferno_amd.engine import NativePopulationLikelihood  # noqa: E402
from gwinferno_amd.lazy import where_finite  # noqa: E402
from gwinferno_amd.synthetic import make_catalog  # noqa: E402


pos = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and not sys.argv[i - 1].startswith("--")]
n_ev, n_pe, n_inj = (int(x) for x in pos[:3]) if len(pos) >= 3 else (30, 2000, 30_000)
pe, inj, total = make_catalog(n_ev, n_pe, n_inj, seed=77)
mmin, mmax, NT = 5.0, 100.0, 10
start = dict(alpha=-2.5, beta=1.0, mpp=35.0, sigpp=5.0, lam=0.1, alpha_a1=2.0, beta_a1=4.0, alpha_a2=2.2, beta_a2=4.4, t_coefs=np.zeros(NT), lamb=2.7)
tilt_model = M.BSplineIIDSpinTilts(NT, pe["cos_tilt_1"], pe["cos_tilt_2"], inj["cos_tilt_1"], inj["cos_tilt_2"], normalize=True)
z_model = M.PowerlawRedshiftModel(pe["redshift"], inj["redshift"])


def weights(d, p, pe_samples):  # what a user writes: a product of the reference's model calls, divided by the sampling prior
    mass = M.plpeak_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], p["alpha"], p["beta"], mmin, mmax, p["mpp"], p["sigpp"], p["lam"])
    mags = M.independent_spin_magnitude_beta_dist(d["a_1"], d["a_2"], p["alpha_a1"], p["beta_a1"], p["alpha_a2"], p["beta_a2"])
    return where_finite(mass * mags * tilt_model(p["t_coefs"], pe_samples=pe_samples) * z_model(d["redshift"], p["lamb"]) / d["prior"])


t0 = time.perf_counter()
wp = weights(pe, start, True)
eng = NativePopulationLikelihood(wp, weights(inj, start, False), z_model.normalization(start["lamb"]))
print(f"engine in {time.perf_counter() - t0:.2f} s: scan kernel '{eng.scan_kernel_name()}', {eng.jit_info()}, dispatch '{eng.dispatch_info()}'")
theta0 = eng.bound.theta_of(wp)
res = eng.evaluate(theta0, total, min_neff_cut=False)
print(f"log_l = {res.log_likelihood:.6f}, |grad| = {np.linalg.norm(res.grad):.4f}, {eng.n_theta} hyper-parameters")
n_it = 2000
print(f"one chain: {1e6 * eng.selftime(theta0, total, n_iter=n_it, min_neff_cut=False):.1f} us per value-and-gradient evaluation")

# the analytic gradient against central differences of the engine's own value, for three of the twenty hyper-parameters
for k in (0, 7, eng.n_theta - 1):
    h = 1e-5 * max(1.0, abs(theta0[k]))
    e = np.zeros_like(theta0)
    e[k] = h
    fd = (eng.evaluate(theta0 + e, total, min_neff_cut=False, want_grad=False).log_likelihood - eng.evaluate(theta0 - e, total, min_neff_cut=False, want_grad=False).log_likelihood) / (2 * h)
    print(f"  d log_l / d theta[{k}]: analytic {res.grad[k]: .6f}, central difference {fd: .6f}")
# sixteen points per launch (vectorised chains): the chain's batched kernel, compiled in the same go
rng = np.random.default_rng(1)
thetas = theta0 + 0.01 * rng.normal(size=(16, eng.n_theta))
vgb = eng.configure_batch(16, total, min_neff_cut=False)
for _ in range(5):
    vgb(thetas)
t0 = time.perf_counter()
for _ in range(100):
    values, grads = vgb(thetas)
dt = time.perf_counter() - t0
print(f"16 points per launch ('{eng.batch_path(16)}'): {1e6 * dt / 1600:.2f} us per evaluation; log_l[0] = {values[0]:.6f}")
eng.close()
