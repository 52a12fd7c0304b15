"""Seeded synthetic catalogs (PE posterior samples + found injections) shaped like the
reference's ``pedict`` / ``injdict`` (pipeline/utils.py:82-96): every PE array is
``(N_events, N_pe)``, every injection array ``(N_inj,)``, keys ``mass_1, mass_ratio, redshift,
a_1, a_2, cos_tilt_1, cos_tilt_2, prior``.

Recipe = SURVEY.md section 8(d).  Used by bench.py, smoke(), the parity tests and the golden
generator, so the GPU path, the oracle and the reference all see byte-identical inputs.
"""
import numpy as np

from .cosmology import planck15_lvk

BASE_SEED = 20250523

# name -> (config id used for the seed, N_ev, N_pe, N_inj)
CONFIG_SIZES = {
    "c1": (1, 10, 1000, 5000),
    "c2": (2, 69, 5000, 50_000),
    "c3": (3, 69, 5000, 100_000),
    "c5": (5, 200, 10_000, 500_000),
}


def _powerlaw_draw(rng, alpha, lo, hi, size):
    """Inverse-CDF draw from x^alpha on [lo, hi] (lo may be an array)."""
    u = rng.uniform(size=size)
    if alpha == -1:
        return lo * np.exp(u * np.log(hi / lo))
    a1 = alpha + 1.0
    return (lo**a1 + u * (hi**a1 - lo**a1)) ** (1.0 / a1)


def make_catalog(n_ev, n_pe, n_inj, seed=BASE_SEED, mmin=5.0, mmax=100.0, zmax_draw=1.9):
    """Return ``(pedict, injdict, total_inj)`` of float64 arrays."""
    rng = np.random.default_rng(seed)
    cosmo = planck15_lvk()

    # ---- PE samples -------------------------------------------------------------------------
    mbar = rng.uniform(8.0, 80.0, size=(n_ev, 1))
    zbar = rng.uniform(0.05, 1.2, size=(n_ev, 1))
    m1 = mbar * rng.lognormal(0.0, 0.15, size=(n_ev, n_pe))
    q = np.clip(rng.beta(5.0, 2.0, size=(n_ev, n_pe)), 0.05, 1.0)
    z = np.clip(zbar * rng.lognormal(0.0, 0.25, size=(n_ev, n_pe)), 1e-3, zmax_draw)
    a1 = rng.uniform(size=(n_ev, n_pe)) ** 1.5
    a2 = rng.uniform(size=(n_ev, n_pe)) ** 1.5
    ct1 = rng.uniform(-1.0, 1.0, size=(n_ev, n_pe))
    ct2 = rng.uniform(-1.0, 1.0, size=(n_ev, n_pe))
    # LVK-style sampling prior: flat in detector-frame component masses, comoving-uniform
    # in redshift (cf. reference preprocess/data_collection.py:122-132)
    prior = cosmo.dVc_dz(z) / (1 + z) * (1 + z) ** 2 * m1 / 4.0
    prior = prior / 1e9  # Gpc^3-ish scale so weights are O(1)
    def derived(m1_, q_, a1_, a2_, ct1_, ct2_):
        """mass_2, chi_eff, chi_p (reference preprocess/conversions.py:1-110 definitions)."""
        chi_eff = (a1_ * ct1_ + q_ * a2_ * ct2_) / (1.0 + q_)
        s1 = a1_ * np.sqrt(np.clip(1.0 - ct1_**2, 0.0, None))
        s2 = a2_ * np.sqrt(np.clip(1.0 - ct2_**2, 0.0, None))
        chi_p = np.maximum(s1, q_ * (4.0 * q_ + 3.0) / (4.0 + 3.0 * q_) * s2)
        return q_ * m1_, chi_eff, chi_p

    m2, chieff, chip = derived(m1, q, a1, a2, ct1, ct2)
    pedict = {
        "mass_2": m2,
        "chi_eff": chieff,
        "chi_p": chip,
        "mass_1": m1,
        "mass_ratio": q,
        "redshift": z,
        "a_1": a1,
        "a_2": a2,
        "cos_tilt_1": ct1,
        "cos_tilt_2": ct2,
        "prior": prior,
    }

    # ---- found injections -------------------------------------------------------------------
    m1i = _powerlaw_draw(rng, -2.35, 2.0, 100.0, n_inj)
    qi = _powerlaw_draw(rng, 1.0, 2.0 / m1i, 1.0, n_inj)
    zg = np.linspace(1e-4, zmax_draw, 4096)
    pz = cosmo.dVc_dz(zg) * (1 + zg)
    cdf = np.concatenate([[0.0], np.cumsum(0.5 * (pz[1:] + pz[:-1]) * np.diff(zg))])
    znorm = cdf[-1]
    zi = np.interp(rng.uniform(size=n_inj), cdf / znorm, zg)
    a1i = rng.uniform(size=n_inj)
    a2i = rng.uniform(size=n_inj)
    ct1i = rng.uniform(-1.0, 1.0, size=n_inj)
    ct2i = rng.uniform(-1.0, 1.0, size=n_inj)
    p_m1 = m1i**-2.35 * (-1.35) / (100.0**-1.35 - 2.0**-1.35)
    p_q = qi * 2.0 / (1.0 - (2.0 / m1i) ** 2)
    p_z = cosmo.dVc_dz(zi) * (1 + zi) / znorm
    prior_i = p_m1 * p_q * p_z * 1.0 * 1.0 * 0.5 * 0.5
    m2i, chieffi, chipi = derived(m1i, qi, a1i, a2i, ct1i, ct2i)
    injdict = {
        "mass_2": m2i,
        "chi_eff": chieffi,
        "chi_p": chipi,
        "mass_1": m1i,
        "mass_ratio": qi,
        "redshift": zi,
        "a_1": a1i,
        "a_2": a2i,
        "cos_tilt_1": ct1i,
        "cos_tilt_2": ct2i,
        "prior": prior_i,
    }
    return pedict, injdict, float(20 * n_inj)


def make_config_catalog(name):
    cid, n_ev, n_pe, n_inj = CONFIG_SIZES[name]
    return make_catalog(n_ev, n_pe, n_inj, seed=BASE_SEED + cid)
