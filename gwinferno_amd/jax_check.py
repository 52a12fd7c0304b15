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
