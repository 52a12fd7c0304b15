#!/usr/bin/env python3
"""The reference's B-spline analysis (examples/simple_bspline_example.py: BASELINE config 5 -- B-spline primary mass
and mass ratio, independent B-spline spin magnitudes and tilts, power law x B-spline redshift) end to end without
JAX / NumPyro: models from the reference's factories (gwinferno_amd.pipeline_utils = gwinferno/pipeline/utils.py:104-160),
its priors (Normal + P-spline smoothing, z_cs[0] pinned to 0: pipeline/utils.py:163-216) handed to the library's C++
NUTS, one chain per engine and host thread.
The likelihood keeps the reference's default cuts (min_neff_cut=True, analysis.py:272-303): on a small synthetic catalog
the flexible model runs into the n_eff wall on most trajectories, which the sampler reports as divergences (as NumPyro
does); `--no-neff-cut` lifts the cut (the setting of the reference's own inference tests, tests/inference_test.py:185).
    python examples/sample_bspline_native.py [n_events n_pe n_inj] [--chains C] [--warmup W] [--samples S] [--no-neff-cut]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gwinferno_amd import pipeline_utils as U  # noqa: E402
from gwinferno_amd.engine import NativePopulationLikelihood  # noqa: E402
from gwinferno_amd.sampling import nuts_engine  # noqa: E402
from gwinferno_amd.synthetic import make_catalog  # noqa: E402


def opt(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


n_chains, n_warm, n_samp = opt("--chains", 2), opt("--warmup", 150), opt("--samples", 150)
pos = [a for i, a in enumerate(sys.argv[1:], 1) if not a.startswith("--") and not sys.argv[i - 1].startswith("--")]
n_ev, n_pe, n_inj = (int(x) for x in pos[:3]) if len(pos) >= 3 else (40, 2000, 40_000)
pe, inj, total = make_catalog(n_ev, n_pe, n_inj, seed=2025)
nspl = {"m1": 30, "q": 14, "a1": 12, "tilt1": 12, "a2": 12, "tilt2": 12, "redshift": 12}  # the example's defaults (BASELINE config 5)
mmin, mmax = 5.0, 100.0
order = ["m1", "q", "a1", "a2", "tilt1", "tilt2", "redshift"]
rng = np.random.default_rng(0)
start = {k: 0.1 * rng.normal(size=nspl[k]) for k in order}
start["redshift"][0] = 0.0
lamb0 = 2.0

engines = []
for c in range(n_chains):  # one engine (its own copy of the catalog on the GPU, its own stream) per chain
    mass_models = U.setup_bspline_mass_models(pe, inj, nspl["m1"], nspl["q"], mmin, mmax)
    mag_model, tilt_model = U.setup_bspline_spin_models(pe, inj, nspl["a1"], nspl["tilt1"], IID=False, a2_nsplines=nspl["a2"], ct2_nsplines=nspl["tilt2"])
    z_model = U.setup_powerlaw_spline_redshift_model(pe, inj, nspl["redshift"])

    def weights(d, flag):  # examples/simple_bspline_example.py:60-71
        return (mass_models(start["m1"], start["q"], pe_samples=flag) * mag_model(start["a1"], start["a2"], pe_samples=flag)
                * tilt_model(start["tilt1"], start["tilt2"], pe_samples=flag) * z_model(d["redshift"], lamb0, start["redshift"]) / d["prior"])

    wp = weights(pe, True)
    engines.append(NativePopulationLikelihood(wp, weights(inj, False), z_model.normalization(lamb0, start["redshift"])))
eng = engines[0]
theta0 = eng.bound.theta_of(wp)


def block(v):  # where a coefficient vector sits in the engine's flat theta
    v = np.atleast_1d(v)
    for off in range(eng.n_theta - len(v) + 1):
        if np.array_equal(theta0[off : off + len(v)], v):
            return slice(off, off + len(v))
    raise RuntimeError("block not found")


slices = {k: block(start[k]) for k in order}
slices["lamb"] = block(lamb0)
prior, bij = U.bspline_example_prior(slices)  # m_tau = q_tau = z_tau = 1, a_tau = ct_tau = 25, lamb ~ Normal(0, 3)
starts = np.stack([theta0 + (0.02 * rng.normal(size=eng.n_theta) if c else 0.0) for c in range(n_chains)])
starts[:, slices["redshift"].start] = 0.0
t0 = time.perf_counter()
res = nuts_engine(engines, total, prior, bij, starts, n_warmup=n_warm, n_samples=n_samp, seed=1, max_tree_depth=8, min_neff_cut="--no-neff-cut" not in sys.argv)
dt = time.perf_counter() - t0
n_lf = sum(r["n_evals"] for r in res)
print(f"{n_ev} events x {n_pe} PE samples, {n_inj} injections, {eng.n_theta} hyper-parameters; {n_chains} chain(s), {n_warm}+{n_samp} iterations: "
      f"{n_lf} likelihood evaluations in {dt:.2f} s ({n_lf / dt:.0f} evals/s)")
for c, r in enumerate(res):
    print(f"  chain {c}: accept {r['accept_rate']:.2f}, step {r['step_size']:.3g}, mean tree depth {r['tree_depth'].mean():.1f}, {r['n_divergent']} divergent")
allth = np.concatenate([r["samples"] for r in res])
print(f"  lamb {allth[:, slices['lamb']].mean():.3f} +- {allth[:, slices['lamb']].std():.3f};  z_cs[0] == 0 in every draw: {bool(np.all(allth[:, slices['redshift'].start] == 0))}")
for k in ("m1", "q"):
    m = allth[:, slices[k]].mean(0)
    print(f"  {k:3s} coefficients (posterior mean): " + " ".join(f"{v:6.2f}" for v in m))
for e in engines:
    e.close()
