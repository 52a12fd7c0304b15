#!/usr/bin/env python3
"""Diagnostic (GPU box): how fast the C++ target (gwi_nuts_engine_lockstep) and the NumPy statement of the same target
(gwi_nuts_run_lockstep + evaluate_batch) drift apart along a chain -- last-bit differences of the prior arithmetic amplified by
the Hamiltonian dynamics, or something else?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_dropin_api import _native_nuts_setup  # noqa: E402
from gwinferno_amd.sampling import nuts_engine_lockstep, nuts_native_lockstep  # noqa: E402

engs, total, prior, bij, theta0, _ = _native_nuts_setup(1)
K = 5
starts = np.stack([theta0 + 0.03 * c for c in range(K)])
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 0
kw = dict(n_warmup=nw, n_samples=70 - nw, seed=11, max_tree_depth=5)
res = nuts_engine_lockstep(engs, K, total, prior, bij, starts, min_neff_cut=False, **kw)
again = nuts_engine_lockstep(engs, K, total, prior, bij, starts, min_neff_cut=False, **kw)
print("engine lock step twice: identical", all(np.array_equal(a["samples"], b["samples"]) for a, b in zip(res, again)))


def batch_target(us, ids):
    fw = [bij.forward(u) for u in us]
    out = engs[0].evaluate_batch(np.stack([f[0] for f in fw]), total, min_neff_cut=False)
    lps, grads = [], []
    for (theta, dth, dlogj, logj), r in zip(fw, out):
        lp, gp = prior(theta)
        lps.append(r.log_likelihood + lp + logj)
        grads.append((r.grad + gp) * dth + dlogj)
    return np.array(lps), np.stack(grads)


ref = nuts_native_lockstep(batch_target, np.stack([bij.inverse(t) for t in starts]), **kw)
for j in range(K):
    th_b = np.array([bij.forward(u)[0] for u in ref[j]["samples"]])
    dev = np.max(np.abs(res[j]["samples"] - th_b) / np.maximum(np.abs(th_b), 1e-3), axis=1)
    print(j, res[j]["n_evals"], ref[j]["n_evals"], f"step {res[j]['step_size']:.3g}", " ".join(f"{d:.0e}" for d in dev[::5]))
