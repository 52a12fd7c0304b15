"""Run ON THE GPU BOX: the library's C++ NUTS on a benchmark configuration under the reference's priors, the draws kept.

    python tools/nuts_trace.py --config c5 --chains 2 --warmup 300 --samples 200 --out gpurun_out/r4/nuts_c5.npz

Writes samples[chains, draws, dim], log_prob[chains, draws], tree_depth[chains, draws], step sizes, and prints the per-chain
(within-chain) and the multi-chain bulk ESS and the split R-hat, to tell "the chains have not met yet" from "the chains mix
slowly" when bench.py's native_nuts line reports a small multi-chain ESS.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import bench
    from gwinferno_amd.compositions import COMPOSITIONS, draw_params
    from gwinferno_amd.sampling import effective_sample_size, nuts_engine, split_rhat
    from gwinferno_amd.synthetic import make_config_catalog

    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c5")
    ap.add_argument("--chains", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--samples", type=int, default=200)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--same-start", action="store_true", help="every chain starts from the first draw (differences are then the sampler's)")
    ap.add_argument("--out", default="gpurun_out/nuts_trace.npz")
    a = ap.parse_args()
    comp_name, cat_name, _, desc = bench.CONFIGS[a.config]
    pe, inj, total = make_config_catalog(cat_name)
    comps = [COMPOSITIONS[comp_name](pe, inj) for _ in range(a.chains)]
    engs = [c.engine(device=0) for c in comps]
    rng = np.random.default_rng(1234)
    thetas = [comps[0].theta(draw_params(comp_name, rng)) for _ in range(64)]
    prior, bij, what = bench.reference_priors(comp_name, comps[0], engs[0].n_theta)
    starts = np.stack([thetas[0]] * a.chains if a.same_start else thetas[:a.chains])
    if bij is not None:
        for k in np.flatnonzero(bij.kind == 3):
            starts[:, k] = bij.lo[k]
    t0 = time.perf_counter()
    res = nuts_engine(engs, total, prior, bij, starts, n_warmup=a.warmup, n_samples=a.samples, max_tree_depth=a.depth, seed=1, min_neff_cut=False)
    dt = time.perf_counter() - t0
    x = np.stack([r["samples"] for r in res])
    lp = np.stack([r["log_prob"] for r in res]) if "log_prob" in res[0] else np.zeros(x.shape[:2])
    depth = np.stack([r["tree_depth"] for r in res])
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    np.savez_compressed(a.out, samples=x, log_prob=lp, tree_depth=depth, step_size=np.array([r["step_size"] for r in res]), starts=starts)
    ess_all = effective_sample_size(x)
    free = np.isfinite(ess_all)
    per_chain = np.stack([effective_sample_size(x[c]) for c in range(a.chains)])
    rh = split_rhat(x)
    rep = {"config": a.config, "workload": desc, "priors": what, "chains": a.chains, "warmup": a.warmup, "samples": a.samples, "wall_s": dt,
           "evals": int(sum(r["n_evals"] for r in res)), "mean_tree_depth": float(depth.mean()), "step_size": [float(r["step_size"]) for r in res],
           "divergences": int(sum(r["n_divergent"] for r in res)), "accept": [float(r["accept_rate"]) for r in res],
           "multi_chain_ess_min_median": [float(np.min(ess_all[free])), float(np.median(ess_all[free]))],
           "within_chain_ess_min_median": [[float(np.nanmin(p)), float(np.nanmedian(p))] for p in per_chain],
           "split_rhat_max_median": [float(np.nanmax(rh[free])), float(np.nanmedian(rh[free]))],
           "log_prob_first_last_mean": [[float(l[:10].mean()), float(l[-10:].mean())] for l in lp]}
    print(json.dumps(rep))
    for e in engs:
        e.close()


if __name__ == "__main__":
    main()
