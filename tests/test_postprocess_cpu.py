"""CPU: host logic of the merger-rate-of-redshift curves (gwinferno_amd/postprocess.py; reference
postprocess/calculations.py:244-276) against the unmodified reference's outputs in tests/golden/ppd_rz.npz.  The engine
is replaced by the NumPy statement of the bound model (tests/bound_eval.py): what is checked here is the term / column /
theta description the functions hand to the engine; the HIP path itself is held to the same golden in
tests/test_gpu_postprocess.py."""
import os

import numpy as np
import pytest

from bound_eval import log_weights

from gwinferno_amd import models as M
from gwinferno_amd import postprocess as P
from gwinferno_amd.engine import bind
from gwinferno_amd.synthetic import make_catalog

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ppd_rz.npz"))


class _BoundOnly:
    """Stand-in for NativePopulationLikelihood: log-weights of the bound model in NumPy."""

    def __init__(self, pe_density, inj_density, hypervolume=None, **kw):
        self.bound = bind(pe_density, inj_density, hypervolume)

    def log_weights(self, theta):
        lpe, linj, _ = log_weights(self.bound, theta)
        return lpe, linj

    def close(self):
        pass


@pytest.fixture()
def z_data(monkeypatch):
    monkeypatch.setattr(P, "NativePopulationLikelihood", _BoundOnly)
    pe, inj, _ = make_catalog(*[int(v) for v in GOLD["catalog"]])
    return pe["redshift"], inj["redshift"]


def _close(a, b, rtol=1e-12):
    return np.all(np.abs(np.asarray(a) - b) <= rtol * np.max(np.abs(b), axis=-1, keepdims=True))


def test_powerlaw_rate_of_z(z_data):
    zm = M.PowerlawRedshiftModel(*z_data)
    rs, zs = P.calculate_powerlaw_rate_of_z_ppds(GOLD["lamb"], GOLD["rate"], zm, pop_frac=GOLD["pop_frac"])
    assert np.array_equal(zs, GOLD["powerlaw/zs"]) and rs.shape == GOLD["powerlaw/rs"].shape
    assert _close(rs, GOLD["powerlaw/rs"])
    rs, _ = P.calculate_powerlaw_rate_of_z_ppds(GOLD["lamb"], GOLD["rate"], zm)  # pop_frac defaults to ones (:246-247)
    assert _close(rs, GOLD["powerlaw/rs_default_frac"])


def test_powerlaw_spline_rate_of_z(z_data):
    zm = M.PowerlawSplineRedshiftModel(int(GOLD["n_splines"]), *z_data)
    rs, zs = P.calculate_powerlaw_spline_rate_of_z_ppds(GOLD["lamb"], GOLD["z_cs"], GOLD["rate"], zm, pop_frac=GOLD["pop_frac"])
    assert np.array_equal(zs, GOLD["spline/zs"])
    assert _close(rs, GOLD["spline/rs"])
    with pytest.raises(ValueError):
        P.calculate_powerlaw_spline_rate_of_z_ppds(GOLD["lamb"], GOLD["z_cs"][:, :-1], GOLD["rate"], zm)
