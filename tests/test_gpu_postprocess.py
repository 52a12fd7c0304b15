"""GPU (-m gpu): posterior-predictive curves (gwinferno_amd/postprocess.py, computed by the likelihood engine
on a mesh "catalog") against the unmodified reference functions' outputs (postprocess/calculations.py:20-276)
stored in tests/golden/ppd.npz and ppd_rz.npz."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ppd.npz"))


def _close(a, b, rtol=1e-9):
    a, b = np.asarray(a), np.asarray(b)
    scale = np.max(np.abs(b), axis=-1, keepdims=True)
    return np.all(np.abs(a - b) <= rtol * scale)


def test_powerlaw_peak_mass_ppds():
    from gwinferno_amd import postprocess as P

    g = {k[len("plpeak_in/"):]: GOLD[k] for k in GOLD.files if k.startswith("plpeak_in/")}
    mp, ms, qp, qs = P.calculate_powerlaw_peak_mass_ppds(g["alpha"], g["beta"], g["mu_peak"], g["sig_peak"], g["lamb"], 5.0, 100.0, rate=GOLD["rate"], pop_frac=GOLD["pop_frac"])
    assert np.array_equal(ms, GOLD["plpeak_out/ms"]) and np.array_equal(qs, GOLD["plpeak_out/qs"])
    assert _close(mp, GOLD["plpeak_out/mpdfs"]) and _close(qp, GOLD["plpeak_out/qpdfs"])
    # each curve integrates to rate * pop_frac
    assert np.allclose(np.trapezoid(mp, ms, axis=1), GOLD["rate"] * GOLD["pop_frac"], rtol=1e-12)


def test_bspline_mass_ppds():
    from gwinferno_amd import postprocess as P

    mp, ms, qp, qs = P.calculate_bspline_mass_ppds(GOLD["bspline_in/m_cs"], GOLD["bspline_in/q_cs"], {"m1": 14, "q": 8}, 5.0, 100.0)
    assert _close(mp, GOLD["bspline_out/mpdfs"]) and _close(qp, GOLD["bspline_out/qpdfs"])


def test_peak_logm1_bspline_q_ppds():
    from gwinferno_amd import postprocess as P

    mp, ms, qp, qs = P.calculate_peak_logm1_bspline_q_ppds(GOLD["peaklog_in/logmp"], GOLD["peaklog_in/logsigp"], GOLD["peaklog_in/q_cs"], {"q": 8}, 5.0, 100.0)
    assert _close(mp, GOLD["peaklog_out/mpdfs"]) and _close(qp, GOLD["peaklog_out/qpdfs"])


def test_one_dimensional_spin_curves():
    from gwinferno_amd import postprocess as P

    ap, aa = P.calculate_beta_spin_mag(GOLD["beta_in/alpha"], GOLD["beta_in/beta"], rate=GOLD["rate"], pop_frac=GOLD["pop_frac"])
    assert np.array_equal(aa, GOLD["beta_out/aa"]) and _close(ap, GOLD["beta_out/apdfs"])
    cp, ct = P.calculate_mixture_iso_aligned_spin_tilt(GOLD["tilt_in/sig"], GOLD["tilt_in/lam"])
    assert np.array_equal(ct, GOLD["tilt_out/ct"]) and _close(cp, GOLD["tilt_out/ctpdfs"])
    ap, aa, cp, cc = P.calculate_bspline_spin_ppds(GOLD["spin_in/a_cs"], GOLD["spin_in/t_cs"], {"a": 10, "tilt": 9})
    assert _close(ap, GOLD["spin_out/apdfs"]) and _close(cp, GOLD["spin_out/ctpdfs"])


def test_rate_of_z_curves():
    """calculations.py:244-276: R(z) = rate pop_frac (1 + z)^lamb [exp(spline(log z))] on the redshift model's own grid."""
    from gwinferno_amd import models as M
    from gwinferno_amd import postprocess as P
    from gwinferno_amd.synthetic import make_catalog

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ppd_rz.npz"))
    pe, inj, _ = make_catalog(*[int(v) for v in g["catalog"]])
    zm = M.PowerlawRedshiftModel(pe["redshift"], inj["redshift"])
    rs, zs = P.calculate_powerlaw_rate_of_z_ppds(g["lamb"], g["rate"], zm, pop_frac=g["pop_frac"])
    assert np.array_equal(zs, g["powerlaw/zs"]) and _close(rs, g["powerlaw/rs"])
    rs, _ = P.calculate_powerlaw_rate_of_z_ppds(g["lamb"], g["rate"], zm)
    assert _close(rs, g["powerlaw/rs_default_frac"])
    zs_model = M.PowerlawSplineRedshiftModel(int(g["n_splines"]), pe["redshift"], inj["redshift"])
    rs, zs = P.calculate_powerlaw_spline_rate_of_z_ppds(g["lamb"], g["z_cs"], g["rate"], zs_model, pop_frac=g["pop_frac"])
    assert np.array_equal(zs, g["spline/zs"]) and _close(rs, g["spline/rs"])
