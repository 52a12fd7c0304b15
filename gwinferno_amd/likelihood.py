"""Drop-in for the reference's likelihood assembly, ``gwinferno/pipeline/analysis.py``.

``hierarchical_likelihood`` keeps the reference's name, positional order, keyword names and defaults
(analysis.py:139-163), registers the same deterministic sites (:260-317) and the same
``numpyro.factor("log_likelihood", ...)`` (:319), and returns the merger rate (:356) -- but takes the
*lazy* PE / injection products built by :mod:`gwinferno_amd.models` and evaluates value, gradient and
all diagnostics in one pass of the HIP engine instead of materialising ``(N_ev, N_pe)`` arrays.

Three ways to drive it
  * NumPyro present (GPU box with jax+numpyro): call it inside a NumPyro model function exactly as
    the reference's ``tests/inference_test.py:162-197`` does; hyper-parameters may be JAX tracers --
    the engine is wrapped in a ``jax.custom_vjp`` around ``jax.pure_callback`` so that
    ``jit(value_and_grad(potential_fn))`` (what NUTS runs) works unchanged.
  * No NumPyro: the same call with concrete numbers records its sites in :func:`last_sites`.
  * Framework-free: :class:`gwinferno_amd.engine.NativePopulationLikelihood` ``.evaluate(theta)``.
The posterior-predictive-check branch (analysis.py:321-355) draws its per-event sample indices on the host from the
engine's per-sample weights (``gwi_log_weights``).  Out of scope: the categorical / mixture
branch (analysis.py:246-254) raises NotImplementedError.
"""
import os

import numpy as np

from .engine import NEG_BIG, NativePopulationLikelihood, structure_key
from .lazy import INJ, PE, Column, Density, Factor, LazyNorm, LogValues, side_of, static_key, static_log_expr

_ENGINES = {}
_LAST_SITES = {}
SAMPLE_VALUES = {}  # values for numpyro.sample sites when numpyro is absent (e.g. {"unscaled_rate": 30.0})


_NUMPYRO = []  # [module | None] once probed: a failing import walks sys.path every time (76 us per model call, measured)


def _numpyro():
    if not _NUMPYRO:
        try:
            import numpyro  # noqa: F401

            _NUMPYRO.append(numpyro)
        except Exception:
            _NUMPYRO.append(None)
    return _NUMPYRO[0]


def last_sites():
    """Sites of the most recent call made WITHOUT numpyro (names as analysis.py:260-319)."""
    return dict(_LAST_SITES)


def clear_engine_cache():
    """Forget every cached engine.  Handles are NOT destroyed here: an engine may still be referenced by a caller of
    :func:`engine_for` or captured in a jitted sampler step (the custom_vjp closure of :func:`_evaluate_jax`); its device
    memory is released when the last reference goes (``NativePopulationLikelihood.__del__``)."""
    _ENGINES.clear()
    _ONE_SIDED.clear()
    _INTERNED.clear()


def _is_traced(values):
    for v in values:
        mod = type(v).__module__ or ""
        if mod.startswith("jax") and not isinstance(v, np.ndarray):
            return True
    return False


def _collect_params(bound, density):
    """Hyper-parameter objects of ``density`` in theta-layout order (scalars and coefficient vectors)."""
    out = []
    for fi, what, k, off in bound.layout:
        f = density.factors[fi]
        out.append(f.scalars[k] if what == "scalar" else (f.norm.coefs if what == "norm_coefs" else f.coefs))
    return out


def engine_for(pe_weights, inj_weights, surveyed_hypervolume=None, device=-1):
    """The cached engine for this model structure + catalog (built, i.e. uploaded, on first use)."""
    key = (structure_key(pe_weights, inj_weights), id(surveyed_hypervolume.owner) if isinstance(surveyed_hypervolume, LazyNorm) else None, device)
    hit = _ENGINES.get(key)
    if hit is None:
        eng = NativePopulationLikelihood(pe_weights, inj_weights, surveyed_hypervolume, device=device)
        # The key is made of object identities: the entry keeps the keyed objects (the data arrays through the factors'
        # columns, the model objects through their owners) alive, so that an id can never come to mean other data.  A
        # model function that hands in NEW arrays on every call gets a new engine every call: the cache is bounded
        # (least recently used out first, GWI_ENGINE_CACHE entries, default 8) so that this costs time, not HBM.
        _ENGINES[key] = hit = (eng, pe_weights, inj_weights, surveyed_hypervolume)
        limit = max(1, int(os.environ.get("GWI_ENGINE_CACHE", "8")))
        while len(_ENGINES) > limit:
            _ENGINES.pop(next(iter(_ENGINES)))  # dropped, not closed: whoever still holds the engine keeps a live handle
    else:
        _ENGINES[key] = _ENGINES.pop(key)  # most recently used last
    return hit[0]


def _sites_from_result(res, Nobs, Tobs, unscaled_rate, flags, xp=np):
    s = res["summary"]
    sites = {
        "log_nEff_inj": s["log_nEff_inj"],
        "log_nEffs": res["log_neffs"],
        "logBFs": res["log_bfs"],
        "detection_efficiency": xp.exp(s["log_det_eff"]),
        "variance_log_BFs": res["variances"],
        "variance_log_detection_efficiency": s["variance_log_detection_efficiency"],
    }
    if flags["reconstruct_rate"]:
        sites["surveyed_hypervolume"] = s["surveyed_hypervolume_norm"] / 1.0e9 * Tobs
        if unscaled_rate is not None:
            sites["rate"] = unscaled_rate / sites["detection_efficiency"] / sites["surveyed_hypervolume"]
    sites["selection_factor"] = s["selection_factor"]
    sites["sum_logBFs"] = s["sum_logBFs"]
    sites["log_l"] = s["log_l"]
    if flags["min_neff_cut"]:
        sites["neff_less_Nobs"] = s["log_likelihood"] if not flags["max_variance_cut"] else s["log_l"]
    sites["variance_log_likelihood"] = s["variance_log_likelihood"]
    if flags["max_variance_cut"]:
        sites["variance_less_1"] = s["log_likelihood"]
    return sites


_SUMMARY_FIELDS = ("log_likelihood", "log_l", "sum_logBFs", "selection_factor", "log_det_eff", "log_nEff_inj", "variance_log_detection_efficiency",
                   "variance_log_likelihood", "min_log_nEff", "surveyed_hypervolume_norm", "log_norm_const")


def _evaluate_numpy(eng, theta, total_inj, Nobs, flags, want_grad=True):
    r = eng.evaluate(theta, total_inj, nobs=Nobs, marginalize_selection=flags["marginalize_selection"], min_neff_cut=flags["min_neff_cut"],
                     max_variance_cut=flags["max_variance_cut"], want_grad=want_grad)
    return {"summary": {k: getattr(r.summary, k) for k in _SUMMARY_FIELDS}, "log_bfs": r.log_bfs, "log_neffs": r.log_neffs, "variances": r.variances, "grad": r.grad}


def _host_callback(eng, total_inj, Nobs, flags):
    """The host side of the JAX seam: ``theta -> (summary[n_sum], per_event[3, n_ev], grad[n_theta])`` as NumPy arrays.
    ``theta`` may carry leading batch dimensions -- what ``jax.vmap`` of the model (NumPyro's ``chain_method="vectorized"``,
    ``vmap`` over initial points) hands a ``pure_callback`` declared with ``vmap_method="broadcast_all"``: the points then go
    through ``gwi_eval_batch`` (one set of launches per <= 16 points: the batched kernels) and every output gets the same
    leading dimensions."""
    n_ev, n_theta = eng.n_ev, eng.n_theta
    kw = dict(nobs=Nobs, marginalize_selection=flags["marginalize_selection"], min_neff_cut=flags["min_neff_cut"], max_variance_cut=flags["max_variance_cut"])
    if getattr(type(eng), "configure_callback", None) is not None:  # (the class's own: a test stand-in that merely forwards attributes takes the generic path below)
        # the lean path: option struct and argument marshalling prepared once, one C call per point / per chunk of <= 64 points,
        # results as three fresh arrays (22 -> ~17 us per leapfrog on config 2, 10 -> ~5 us per point with 16 vectorised chains)
        call, call_batch = eng.configure_callback(total_inj, summary_fields=_SUMMARY_FIELDS, **kw)
        step_lean = min(64, max(1, int(os.environ.get("GWI_MAX_BATCH", "16"))))

        def host_lean(theta):
            theta = np.asarray(theta, dtype=np.float64)
            if theta.ndim == 1:
                return call(theta)
            lead = theta.shape[:-1]
            flat = np.ascontiguousarray(theta.reshape(-1, n_theta))
            parts = [call_batch(flat[i : i + step_lean]) for i in range(0, flat.shape[0], step_lean)]
            summ, per_event, grad = (np.concatenate([p[j] for p in parts]) for j in range(3)) if len(parts) > 1 else parts[0]
            return summ.reshape(lead + (len(_SUMMARY_FIELDS),)), per_event.reshape(lead + (3, n_ev)), grad.reshape(lead + (n_theta,))

        return host_lean

    def pack(r):
        summ = np.array([getattr(r.summary, k) for k in _SUMMARY_FIELDS])
        per_event = np.stack([r.log_bfs, r.log_neffs, r.variances])
        return summ, per_event, (r.grad if r.grad is not None else np.zeros(n_theta))

    def host(theta):
        theta = np.asarray(theta, dtype=np.float64)
        if theta.ndim == 1:
            return pack(eng.evaluate(theta, total_inj, want_grad=True, **kw))
        lead = theta.shape[:-1]
        flat = theta.reshape(-1, n_theta)
        step = min(64, max(1, int(os.environ.get("GWI_MAX_BATCH", "16"))))  # the engine clamps max_batch to [1, 64] (gwi_create)
        rows = []
        for i in range(0, flat.shape[0], step):
            rows += [pack(r) for r in eng.evaluate_batch(flat[i : i + step], total_inj, want_grad=True, **kw)]
        return (np.stack([r[0] for r in rows]).reshape(lead + (len(_SUMMARY_FIELDS),)), np.stack([r[1] for r in rows]).reshape(lead + (3, n_ev)),
                np.stack([r[2] for r in rows]).reshape(lead + (n_theta,)))

    return host


_WARNED_F32 = []


def _jax_dtypes(jax):
    """``(float dtype, int dtype)`` that cross the JAX seam: float64 / int64 where ``jax_enable_x64`` is on, else JAX's
    defaults float32 / int32.  The reference runs in whatever precision JAX is configured for -- fp32 unless the user exports
    ``JAX_ENABLE_X64=1`` -- so the seam follows the configuration instead of insisting on fp64: with x64
    off, hyper-parameters arrive as fp32, the engine evaluates them in fp64 as always, and value, sites and gradient go back
    as fp32 (declaring float64 results to ``pure_callback`` under x64-off JAX is a dtype error inside the callback, not a
    message).  Says so once."""
    try:
        f = np.dtype(jax.dtypes.canonicalize_dtype(np.float64))
        i = np.dtype(jax.dtypes.canonicalize_dtype(np.int64))
    except Exception:  # a JAX without jax.dtypes.canonicalize_dtype: read the switch
        on = bool(getattr(jax.config, "jax_enable_x64", False))
        f, i = (np.dtype(np.float64), np.dtype(np.int64)) if on else (np.dtype(np.float32), np.dtype(np.int32))
    if f != np.float64 and not _WARNED_F32:
        import warnings

        _WARNED_F32.append(True)
        warnings.warn("gwinferno_amd: JAX runs without jax_enable_x64, so hyper-parameters, sites and gradients cross the NumPyro seam in float32 "
                      "(the engine itself computes in float64).  Export JAX_ENABLE_X64=1 (or jax.config.update('jax_enable_x64', True)) for the "
                      "fp64 behaviour the <= 1e-9 parity figures refer to.", RuntimeWarning, stacklevel=3)
    return f, i


def _evaluate_jax(eng, params, total_inj, Nobs, flags):
    """``jax.custom_vjp`` over ``jax.pure_callback``: value and every site from one engine call, the
    gradient handed to JAX's reverse mode.  (Exercised only where jax is installed.)"""
    import jax
    import jax.numpy as jnp

    n_ev, n_theta, n_sum = eng.n_ev, eng.n_theta, len(_SUMMARY_FIELDS)
    fdt, _ = _jax_dtypes(jax)
    shapes = (
        jax.ShapeDtypeStruct((n_sum,), fdt),
        jax.ShapeDtypeStruct((3, n_ev), fdt),
        jax.ShapeDtypeStruct((n_theta,), fdt),
    )
    host64 = _host_callback(eng, total_inj, Nobs, flags)
    if fdt == np.float64:
        host = host64
    else:  # x64 off: the engine's float64 results leave in the dtype JAX was told to expect (nan_to_num(-inf) becomes float32's)

        def narrow(o):
            o = np.asarray(o, dtype=np.float64)
            with np.errstate(over="ignore"):
                out = o.astype(fdt)
            big = np.isfinite(o) & ~np.isfinite(out)  # finite in float64 (the cuts' nan_to_num(-inf)), beyond float32: its extreme value, as jnp.nan_to_num gives there
            out[big] = np.sign(o[big]) * np.finfo(fdt).max
            return out

        def host(theta):
            return tuple(narrow(o) for o in host64(theta))

    # batched under vmap (vectorised chains): one call with the batch in front, see _host_callback.  Whether this JAX knows
    # `vmap_method` is read from the signature once -- a try / except around the call would also swallow unrelated TypeErrors
    import inspect

    try:
        has_vmap_method = "vmap_method" in inspect.signature(jax.pure_callback).parameters
    except (TypeError, ValueError):
        has_vmap_method = False
    cb_kw = {"vmap_method": "broadcast_all"} if has_vmap_method else {}

    def callback(theta):
        return jax.pure_callback(host, shapes, theta, **cb_kw)

    @jax.custom_vjp
    def f(theta):
        summ, per_event, _ = callback(theta)
        return summ, per_event

    def f_fwd(theta):
        summ, per_event, grad = callback(theta)
        return (summ, per_event), grad

    def f_bwd(grad, cts):
        ct_summ, _ = cts
        # only log_likelihood (summary[0]) is differentiable: it is what numpyro.factor consumes
        return (ct_summ[0] * grad,)

    f.defvjp(f_fwd, f_bwd)
    theta = jnp.concatenate([jnp.ravel(jnp.asarray(p, dtype=fdt)) for p in params])
    summ, per_event = f(theta)
    summary = {k: summ[i] for i, k in enumerate(_SUMMARY_FIELDS)}
    return {"summary": summary, "log_bfs": per_event[0], "log_neffs": per_event[1], "variances": per_event[2], "grad": None}


def _ppc_indices(eng, theta, pedata, injdata, n_obs, m1min, m2min, mmax):
    """analysis.py:321-344 on the host: per-sample weights from the engine (``gwi_log_weights``), the reference's mass
    cuts, then one index per event from the PE weights and one from the injection weights.  The reference draws with
    ``jax.random.choice`` keyed by ``PRNGKey(ev)``; here the stream is ``numpy.random.default_rng([ev, 0 | 1])`` and the
    draw an inverse-CDF lookup -- the same distribution, not JAX's threefry bits.  Returns int64 ``(2, n_obs)``."""
    lw_pe, lw_inj = eng.log_weights(theta)
    with np.errstate(all="ignore"):
        m1, q = np.asarray(pedata["mass_1"]), np.asarray(pedata["mass_ratio"])
        w_pe = np.exp(lw_pe - np.max(lw_pe, axis=1, keepdims=True))
        w_pe = np.where((m1 < m1min) | (m1 > mmax) | (m1 * q < m2min) | ~np.isfinite(w_pe), 0.0, w_pe)
        m1i, qi = np.asarray(injdata["mass_1"]), np.asarray(injdata["mass_ratio"])
        w_inj = np.exp(lw_inj - np.max(lw_inj))
        w_inj = np.where((m1i < m1min) | (m1i > mmax) | (m1i * qi < m2min) | ~np.isfinite(w_inj), 0.0, w_inj)
    out = np.zeros((2, n_obs), dtype=np.int64)
    cdf_inj = np.cumsum(w_inj)
    for ev in range(n_obs):
        cdf = np.cumsum(w_pe[ev])
        out[0, ev] = min(int(np.searchsorted(cdf, np.random.default_rng([ev, 0]).uniform() * cdf[-1], side="right")), cdf.size - 1)
        out[1, ev] = min(int(np.searchsorted(cdf_inj, np.random.default_rng([ev, 1]).uniform() * cdf_inj[-1], side="right")), cdf_inj.size - 1)
    return out


def _posterior_predictive_sites(eng, params, n_obs, param_names, pedata, injdata, m1min, m2min, mmax, traced):
    """The ``{p}_obs_event_{ev}`` / ``{p}_pred_event_{ev}`` sites of analysis.py:350-355."""
    if traced:
        import jax
        import jax.numpy as jnp

        fdt, idt = _jax_dtypes(jax)
        theta = jnp.concatenate([jnp.ravel(jnp.asarray(p, dtype=fdt)) for p in params])
        idx = jax.pure_callback(lambda th: _ppc_indices(eng, np.asarray(th, dtype=np.float64), pedata, injdata, n_obs, m1min, m2min, mmax).astype(idt),
                                jax.ShapeDtypeStruct((2, n_obs), idt), theta)
        take_pe = lambda p, ev: jnp.asarray(pedata[p])[ev, idx[0, ev]]  # noqa: E731
        take_inj = lambda p, ev: jnp.asarray(injdata[p])[idx[1, ev]]  # noqa: E731
    else:
        theta = np.concatenate([np.ravel(np.asarray(p, dtype=np.float64)) for p in params])
        idx = _ppc_indices(eng, theta, pedata, injdata, n_obs, m1min, m2min, mmax)
        take_pe = lambda p, ev: np.asarray(pedata[p])[ev, idx[0, ev]]  # noqa: E731
        take_inj = lambda p, ev: np.asarray(injdata[p])[idx[1, ev]]  # noqa: E731
    sites = {}
    for ev in range(n_obs):
        for p in param_names:
            sites[f"{p}_obs_event_{ev}"] = take_pe(p, ev)
            sites[f"{p}_pred_event_{ev}"] = take_inj(p, ev)
    return sites


def hierarchical_likelihood(
    pe_weights,
    inj_weights,
    total_inj,
    Nobs,
    Tobs,
    surveyed_hypervolume=None,
    categorical=False,
    marginal_qs=False,
    indv_weights=None,
    rngkey=None,
    pop_frac=None,
    reconstruct_rate=True,
    marginalize_selection=False,
    min_neff_cut=True,
    max_variance_cut=False,
    posterior_predictive_check=False,
    param_names=None,
    pedata=None,
    injdata=None,
    m2min=3.0,
    m1min=5.0,
    mmax=100.0,
    log=False,
):
    """Same contract as the reference (analysis.py:139-356); ``pe_weights`` / ``inj_weights`` are lazy
    densities.  ``log`` is accepted for signature compatibility: the engine works in the log domain
    with an online maximum either way, so the linear and log forms of the reference coincide."""
    if max_variance_cut and (marginalize_selection or min_neff_cut):
        raise ValueError(
            "max_variance_cut is True which requires marginalize_selection and "
            "min_neff_cut to be False but got "
            f"marginalize_selection = {marginalize_selection} "
            f"and min_neff_cut = {min_neff_cut}",
        )
    if categorical:
        raise NotImplementedError("categorical sub-population assignment (analysis.py:246-254) is outside the accelerated path")
    if marginal_qs:
        raise NotImplementedError("marginal_qs belongs to the categorical branch (analysis.py:246-254, :347-349), outside the accelerated path")
    # Plain arrays of weights, as the reference takes them (analysis.py:139-163: `pe_weights` (N_ev, N_pe), `inj_weights`
    # (N_inj,); log-weights with log=True): the same engine on a unit factor (see _array_density).  They carry no
    # hyper-parameters, so there is nothing to differentiate and `surveyed_hypervolume` is the caller's number.
    hypervolume_value = None
    arrays = not isinstance(pe_weights, Density) and not isinstance(inj_weights, Density)
    if arrays:
        if _is_traced([pe_weights, inj_weights]):
            raise TypeError("array-valued pe_weights / inj_weights must be concrete (NumPy) arrays: inside a traced model function build them from "
                            "gwinferno_amd.models, whose lazy products the engine differentiates")
        pe_weights, inj_weights = _array_density(np.asarray(pe_weights), log), _array_density(np.asarray(inj_weights), log)
        if reconstruct_rate:
            if surveyed_hypervolume is None or isinstance(surveyed_hypervolume, LazyNorm):
                raise TypeError("with array-valued weights surveyed_hypervolume must be a number (z_model.normalization evaluated by the caller) when reconstruct_rate=True")
            hypervolume_value, surveyed_hypervolume = float(surveyed_hypervolume), None
    if not isinstance(pe_weights, Density) or not isinstance(inj_weights, Density):
        raise TypeError("pe_weights / inj_weights must both be lazy densities from gwinferno_amd.models, or both arrays of weights")
    if reconstruct_rate and not arrays and not isinstance(surveyed_hypervolume, LazyNorm):
        raise TypeError("surveyed_hypervolume must be z_model.normalization(...) when reconstruct_rate=True")

    flags = dict(marginalize_selection=bool(marginalize_selection), min_neff_cut=bool(min_neff_cut), max_variance_cut=bool(max_variance_cut), reconstruct_rate=bool(reconstruct_rate))
    eng = engine_for(pe_weights, inj_weights, surveyed_hypervolume if isinstance(surveyed_hypervolume, LazyNorm) else None)
    params = _collect_params(eng.bound, pe_weights)
    npro = _numpyro()

    unscaled_rate = None
    if reconstruct_rate:
        if npro is not None:
            import numpyro.distributions as dist

            unscaled_rate = npro.sample("unscaled_rate", dist.Gamma(Nobs))  # analysis.py:268
        else:
            unscaled_rate = SAMPLE_VALUES.get("unscaled_rate")

    if _is_traced(params):
        import jax.numpy as jnp

        res, xp = _evaluate_jax(eng, params, float(total_inj), float(Nobs), flags), jnp
    else:
        theta = np.concatenate([np.ravel(np.asarray(p, dtype=np.float64)) for p in params])
        res, xp = _evaluate_numpy(eng, theta, float(total_inj), float(Nobs), flags), np

    if hypervolume_value is not None:
        res["summary"]["surveyed_hypervolume_norm"] = hypervolume_value
    sites = _sites_from_result(res, Nobs, Tobs, unscaled_rate, flags, xp=xp)
    log_l = res["summary"]["log_likelihood"]
    if posterior_predictive_check and param_names is not None and injdata is not None and pedata is not None:  # analysis.py:320-355
        sites.update(_posterior_predictive_sites(eng, params, int(Nobs), param_names, pedata, injdata, m1min, m2min, mmax, traced=xp is not np))
    if npro is not None:
        for name, value in sites.items():
            npro.deterministic(name, value)
        npro.factor("log_likelihood", log_l)  # analysis.py:319
    else:
        _LAST_SITES.clear()
        _LAST_SITES.update(sites)
        _LAST_SITES["log_likelihood"] = log_l
        if res["grad"] is not None:
            _LAST_SITES["grad_log_likelihood"] = res["grad"]
    return sites.get("rate")


def construct_hierarchical_model(
    model_dict,
    prior_dict,
    marginalize_selection=False,
    min_neff_cut=True,
    max_variance_cut=False,
    posterior_predictive_check=True,
):
    """analysis.py:359-424: the model function ``model(samps, injs, Ninj, Nobs, Tobs)`` assembled from
    ``model_dict[source parameter] -> PopModel`` (or the name of another parameter whose distribution it
    shares, :394-399) and ``prior_dict[hyper-parameter] -> PopPrior | constant``.  Every population model's
    ``log_prob`` is summed per sample, ``- log prior`` (:401-402), and handed to
    :func:`hierarchical_likelihood` with ``log=True`` and the redshift model's ``norm`` as surveyed
    hypervolume (:404-423).

    The distribution classes are those of :mod:`gwinferno_amd.numpyro_distributions` (or anything whose
    ``log_prob`` returns a lazy log-density).  Hyper-parameters with a ``PopPrior`` are drawn with
    ``numpyro.sample`` where NumPyro is installed and read from ``SAMPLE_VALUES[name]`` where it is not.
    Mixture models (``PopMixtureModel``, :383-389) are NumPyro mixture distributions, not part of the
    accelerated path.  The reference's default ``posterior_predictive_check=True`` (analysis.py:321-355) is honoured:
    the per-event draws are made on the host from the engine's per-sample weights (see :func:`_ppc_indices`)."""
    from .lazy import log as lazy_log
    from .parser import PopMixtureModel, PopModel

    source_param_names = [k for k in model_dict.keys()]
    hyper_params = {k: None for k in prior_dict.keys()}
    pop_models = {k: None for k in model_dict.keys()}
    z_grid = None
    if "redshift" in pop_models.keys():
        z_grid = np.linspace(1e-9, prior_dict["redshift_maximum"], 1000)  # :371-372

    def model(samps, injs, Ninj, Nobs, Tobs):
        npro = _numpyro()
        for k, v in prior_dict.items():
            if hasattr(v, "dist") and hasattr(v, "params"):  # :376-379
                hyper_params[k] = npro.sample(k, v.dist(**v.params)) if npro is not None else SAMPLE_VALUES[k]
            else:
                hyper_params[k] = v
        iid_mapping = {}
        for k, v in model_dict.items():
            if isinstance(v, PopMixtureModel):
                raise NotImplementedError("mixture population models (analysis.py:383-389) are outside the accelerated path")
            elif isinstance(v, PopModel):
                hps = {p: hyper_params[f"{k}_{p}"] for p in v.params}
                if k == "redshift":
                    hps["grid"] = z_grid
                pop_models[k] = v.model(**hps)
            elif isinstance(v, str):
                iid_mapping[v] = k
            else:
                raise ValueError(f"Unknown model type: {type(v)}:{v}")
        for shared_param, param in iid_mapping.items():
            pop_models[shared_param] = pop_models[param]

        inj_weights = sum(pop_models[k].log_prob(injs[k]) for k in source_param_names) - lazy_log(injs["prior"])
        pe_weights = sum(pop_models[k].log_prob(samps[k]) for k in source_param_names) - lazy_log(samps["prior"])

        return hierarchical_likelihood(
            pe_weights,
            inj_weights,
            total_inj=Ninj,
            Nobs=Nobs,
            Tobs=Tobs,
            surveyed_hypervolume=pop_models["redshift"].norm,
            marginalize_selection=marginalize_selection,
            min_neff_cut=min_neff_cut,
            max_variance_cut=max_variance_cut,
            posterior_predictive_check=posterior_predictive_check,
            pedata=samps,
            injdata=injs,
            param_names=source_param_names,
            m1min=2.0,
            m2min=2.0,
            mmax=100.0,
            log=True,
        )

    return model


_ONE_SIDED = {}
_ZEROS = {}     # shape -> the one zeros column array-valued weights of that shape share (_array_density)
_INTERNED = {}  # (shape, content hash) -> the first array seen with that content


def _cut(arr, side):
    """This side's array -> a small stand-in for the OTHER side (first event's samples as a 1-D
    'injection' set; the first <= 256 injections as a one-event PE tensor)."""
    a = np.asarray(arr)
    if side == PE:
        return np.ascontiguousarray(a[0])
    return np.ascontiguousarray(a[None, : min(a.shape[0], 256)])


def _mirror(density, side):
    """A structure-identical density for the other side, cut from ``density``'s own data.  The engine
    always scans a PE tensor and an injection set together; for the one-sided reference functions
    below the mirrored side is ballast (a few hundred samples) whose results are discarded."""
    other = INJ if side == PE else PE
    memo = {}
    cut = lambda e: None if e is None else e.substitute(lambda a: _cut(a, side), memo)  # noqa: E731 -- shared sources stay shared
    factors = []
    for f in density.factors:
        factors.append(Factor(f.kind, other, [Column("id", cut(c.expr())) for c in f.columns], scalars=f.scalars, coefs=f.coefs, consts=f.consts,
                              n_basis=f.n_basis, flags=f.flags, mask=cut(f.mask_expr()), static_log=cut(f.static_log_expr()), norm=f.norm, owner=f.owner,
                              norm_owner=f.norm_owner, tag=f.tag))
    return Density(factors, other, [(sgn, LogValues(expr=cut(static_log_expr(a)))) for sgn, a in density.log_static], density.log_const)


def _array_density(weights, log):
    """A plain array of importance weights (what the reference's two functions take, analysis.py:50-136) as a lazy density:
    ``exp(kappa)`` with ``kappa = log(weights)`` (``weights`` itself when ``log``) times a unit factor -- a bare power law
    ``x^0`` on a column of zeros (the engine's one-term chain "pl"), so that the same scan, the same online-maximum
    reductions and the same record assembly serve arrays as serve models.  Zero weights (``log``: ``-inf``) are excluded
    samples, as they contribute nothing to the reference's sums; NaN / negative weights count as zero like every NaN weight
    on this path (tests/inference_test.py:172)."""
    from . import _native as N

    w = np.ascontiguousarray(weights, dtype=np.float64)
    side = side_of(w)
    # The engine caches key columns and static factors by IDENTITY of their source arrays (lazy.Column.key, static_key); the
    # reference calls these functions once per likelihood evaluation with a freshly computed array, so equal arrays must map to
    # the same objects here or every call would build (and evict) an engine: one shared zeros column per shape, and the weights
    # themselves interned by a content hash (one pass over the data, as LogValues(values=...) does for log-weights).
    zeros = _ZEROS.get(w.shape)
    if zeros is None:
        zeros = _ZEROS[w.shape] = np.zeros(w.shape)
        while len(_ZEROS) > 8:
            _ZEROS.pop(next(iter(_ZEROS)))
    if not log:
        import hashlib

        digest = (w.shape, hashlib.blake2b(memoryview(w).cast("B"), digest_size=16).hexdigest())
        w = _INTERNED.setdefault(digest, w)
        _INTERNED[digest] = _INTERNED.pop(digest)  # most recently used last
        while len(_INTERNED) > max(1, int(os.environ.get("GWI_ENGINE_CACHE", "8"))):
            _INTERNED.pop(next(iter(_INTERNED)))
    unit = Factor(N.TERM_POWERLAW, side, [Column("id", zeros)], [0.0], consts=(0.0, 1.0), flags=N.POWERLAW_UNNORMALISED, tag="array-weights")
    return Density([unit], side, [(1.0, LogValues(values=w) if log else w)], 0.0)


def _one_sided(weights, log=False):
    if not isinstance(weights, Density):
        if isinstance(weights, (np.ndarray, list, tuple)) or (hasattr(weights, "__array__") and np.ndim(weights) > 0):
            weights = _array_density(np.asarray(weights), log)
        else:
            raise TypeError("weights must be a lazy density from gwinferno_amd.models or an array of importance weights")
    if not weights.factors:
        raise TypeError("weights must be a lazy density from gwinferno_amd.models or an array of importance weights")
    side = weights.side if weights.side is not None else side_of(weights.factors[0].columns[0].expr())
    key = (side, tuple(f.structure() + tuple(c.key() for c in f.columns) for f in weights.factors), tuple((sgn, static_key(a)) for sgn, a in weights.log_static))
    hit = _ONE_SIDED.get(key)
    if hit is None:
        mirror = _mirror(weights, side)
        eng = NativePopulationLikelihood(weights, mirror) if side == PE else NativePopulationLikelihood(mirror, weights)
        hit = _ONE_SIDED[key] = (mirror, eng, weights)  # the originals stay referenced: keys are object ids
        while len(_ONE_SIDED) > max(1, int(os.environ.get("GWI_ENGINE_CACHE", "8"))):  # bounded like the two-sided cache
            _ONE_SIDED.pop(next(iter(_ONE_SIDED)))
    eng = hit[1]
    theta = np.concatenate([np.ravel(np.asarray(p, dtype=np.float64)) for p in _collect_params(eng.bound, weights)])
    return side, eng, theta


def per_event_log_bayes_factors(weights, log=False):
    """analysis.py:50-88 for a lazy PE product ``(N_ev, N_pe)`` -- or, as in the reference, a plain ``(N_ev, N_pe)`` array of
    weights (``log=True``: of log-weights): returns ``(logBFs, logn_effs, variances)``, each ``(N_ev,)``.  For a lazy product
    ``log`` only matters for the signature (the engine works in the log domain with an online maximum, where the reference's two
    branches coincide).  Inside a likelihood prefer :func:`hierarchical_likelihood`, which produces the same arrays as sites
    from the one fused scan."""
    side, eng, theta = _one_sided(weights, log)
    if side != PE:
        raise ValueError("per_event_log_bayes_factors expects the (N_events, N_samples) PE product")
    r = eng.evaluate(theta, float(eng.n_inj), min_neff_cut=False, want_grad=False)
    return r.log_bfs, r.log_neffs, r.variances


def detection_efficiency(weights, Ninj, log=False):
    """analysis.py:91-136 for a lazy injection product ``(N_found,)`` or a plain array of weights (``log=True``: log-weights):
    returns ``(logmu, logn_eff, variance)``."""
    side, eng, theta = _one_sided(weights, log)
    if side != INJ:
        raise ValueError("detection_efficiency expects the (N_found_injections,) injection product")
    r = eng.evaluate(theta, float(Ninj), min_neff_cut=False, want_grad=False)
    s = r.summary
    return s.log_det_eff, s.log_nEff_inj, s.variance_log_detection_efficiency


__all__ = ["hierarchical_likelihood", "construct_hierarchical_model", "per_event_log_bayes_factors", "detection_efficiency", "last_sites", "engine_for", "clear_engine_cache", "SAMPLE_VALUES", "NEG_BIG"]
