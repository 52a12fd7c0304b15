"""One ``jit(value_and_grad(potential_energy))`` of a NumPyro model through the engine -- what NumPyro's NUTS evaluates
per leapfrog (reference: tests/inference_test.py:313-347; pipeline/analysis.py:260-319).  Needs jax + numpyro, which the
build and test images lack: ``__graft_entry__.smoke()`` calls this only where both import; the adapter's plumbing itself is
covered on the CPU under a shim (tests/test_jax_adapter_cpu.py)."""
import numpy as np


def potential_energy_value_and_grad(pedict, injdict, total_inj, point=None):
    import jax
    import jax.numpy as jnp
    import numpyro
    import numpyro.distributions as dist
    from numpyro.infer.util import potential_energy

    from .lazy import where_finite
    from .likelihood import hierarchical_likelihood
    from .models import PowerlawRedshiftModel, powerlaw_primary_ratio_pdf

    jax.config.update("jax_enable_x64", True)
    n_obs = int(np.asarray(pedict["mass_1"]).shape[0])
    z_model = PowerlawRedshiftModel(z_pe=pedict["redshift"], z_inj=injdict["redshift"])

    def model():  # tests/inference_test.py:162-197 with the drop-in names
        alpha = numpyro.sample("alpha", dist.Normal(0, 2))
        beta = numpyro.sample("beta", dist.Normal(0, 2))
        lamb = numpyro.sample("lamb", dist.Normal(0, 3))

        def get_weights(d):
            return where_finite(powerlaw_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], alpha=alpha, beta=beta, mmin=5.0, mmax=100.0) * z_model(d["redshift"], lamb) / d["prior"])

        hierarchical_likelihood(get_weights(pedict), get_weights(injdict), total_inj=total_inj, Nobs=n_obs, Tobs=1.0, surveyed_hypervolume=z_model.normalization(lamb=lamb),
                                marginalize_selection=False, min_neff_cut=False)

    params = {k: jnp.asarray(v, dtype=jnp.float64) for k, v in (point or {"alpha": -2.3, "beta": 0.8, "lamb": 2.5}).items()}
    params["unscaled_rate"] = jnp.asarray(np.log(float(n_obs)))  # unconstrained (log) value of the Gamma site of analysis.py:268
    f = jax.jit(jax.value_and_grad(lambda p: potential_energy(model, (), {}, p)))
    value, grad = f(params)
    return float(value), np.concatenate([np.ravel(np.asarray(grad[k])) for k in sorted(grad)])


def main(argv=None):
    """``python -m gwinferno_amd.jax_check [--nuts N]``: the one-command check of the NumPyro seam for a box that has JAX,
    NumPyro and an MI355X (neither the build nor the test image has JAX: the result of this command is what a user pastes
    into an issue).  Prints one JSON document:

    * versions, platform;
    * ``potential_energy`` value and gradient of the reference's parametric test model (tests/inference_test.py:162-197)
      through ``jit(value_and_grad(...))`` and the ``custom_vjp`` / ``pure_callback`` adapter (gwinferno_amd/likelihood.py);
    * the same gradient by central differences of the adapter's own value (the check of tests/inference_test.py:320-347 is
      finiteness; this one is stricter) and the engine's direct ``evaluate`` at the same point;
    * with ``--nuts N``: N warm-up + N sampling iterations of ``numpyro.infer.NUTS`` (the reference's sampler,
      examples/utils.py:63-85) driving the engine unchanged, with its acceptance rate and mean tree depth.
    """
    import argparse
    import json
    import platform
    import sys

    ap = argparse.ArgumentParser(prog="python -m gwinferno_amd.jax_check")
    ap.add_argument("--nuts", type=int, default=0, help="also run numpyro NUTS for this many warm-up and this many sampling iterations")
    ap.add_argument("--events", type=int, default=12)
    ap.add_argument("--pe", type=int, default=256)
    ap.add_argument("--inj", type=int, default=4096)
    args = ap.parse_args(argv)
    report = {"python": platform.python_version(), "platform": platform.platform()}
    try:
        import jax
        import numpyro

        report.update(jax=jax.__version__, numpyro=numpyro.__version__, jax_backend=jax.default_backend())
    except Exception as exc:
        report["error"] = f"jax / numpyro not importable: {type(exc).__name__}: {exc}"
        print(json.dumps(report, indent=1))
        return 2
    from .engine import NativePopulationLikelihood  # noqa: F401  (fails loudly without the HIP library / a gfx950 device)
    from .synthetic import make_catalog

    pe, inj, total = make_catalog(args.events, args.pe, args.inj, seed=7)
    point = {"alpha": -2.3, "beta": 0.8, "lamb": 2.5}
    value, grad = potential_energy_value_and_grad(pe, inj, total, point)
    names = sorted(list(point) + ["unscaled_rate"])
    fd = []
    for k in names:
        if k == "unscaled_rate":
            fd.append(None)
            continue
        h = 1e-5
        up, dn = dict(point), dict(point)
        up[k] += h
        dn[k] -= h
        fd.append((potential_energy_value_and_grad(pe, inj, total, up)[0] - potential_energy_value_and_grad(pe, inj, total, dn)[0]) / (2 * h))
    report["potential_energy"] = {"value": value, "parameters": names, "grad": [float(g) for g in grad], "central_difference_grad": fd,
                                  "finite": bool(np.isfinite(value) and np.all(np.isfinite(grad)))}
    errs = [abs(g - f) / max(1.0, abs(f)) for g, f in zip(grad, fd) if f is not None]
    report["potential_energy"]["max_rel_err_vs_differences"] = float(max(errs))
    ok = report["potential_energy"]["finite"] and max(errs) < 1e-5
    if args.nuts > 0:
        import jax
        import jax.numpy as jnp
        import numpyro.distributions as dist
        from numpyro.infer import MCMC, NUTS

        from .lazy import where_finite
        from .likelihood import hierarchical_likelihood
        from .models import PowerlawRedshiftModel, powerlaw_primary_ratio_pdf

        z_model = PowerlawRedshiftModel(z_pe=pe["redshift"], z_inj=inj["redshift"])

        def model():
            alpha = numpyro.sample("alpha", dist.Normal(0, 2))
            beta = numpyro.sample("beta", dist.Normal(0, 2))
            lamb = numpyro.sample("lamb", dist.Normal(0, 3))

            def w(d):
                return where_finite(powerlaw_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], alpha=alpha, beta=beta, mmin=5.0, mmax=100.0) * z_model(d["redshift"], lamb) / d["prior"])

            hierarchical_likelihood(w(pe), w(inj), total_inj=total, Nobs=args.events, Tobs=1.0, surveyed_hypervolume=z_model.normalization(lamb=lamb), min_neff_cut=False)

        mcmc = MCMC(NUTS(model, max_tree_depth=6), num_warmup=args.nuts, num_samples=args.nuts, progress_bar=False)
        mcmc.run(jax.random.PRNGKey(0), extra_fields=("accept_prob", "num_steps"))
        ex = mcmc.get_extra_fields()
        s = mcmc.get_samples()
        report["numpyro_nuts"] = {"iterations": args.nuts, "mean_accept_prob": float(jnp.mean(ex["accept_prob"])), "mean_num_steps": float(jnp.mean(ex["num_steps"])),
                                  "posterior_means": {k: float(jnp.mean(v)) for k, v in s.items()}, "all_finite": bool(all(bool(jnp.all(jnp.isfinite(v))) for v in s.values()))}
        ok = ok and report["numpyro_nuts"]["all_finite"]
    report["ok"] = bool(ok)
    print(json.dumps(report, indent=1))
    return 0 if ok else 1


if __name__ == "__main__":
    import sys

    sys.exit(main())
