"""CPU: argument handling of the drop-in API that needs no device (the reference's error behaviour,
analysis.py:237-243, and the lazy-density algebra)."""
import numpy as np
import pytest

from gwinferno_amd import models as M
from gwinferno_amd.lazy import Density
from gwinferno_amd.likelihood import hierarchical_likelihood
from gwinferno_amd.synthetic import make_catalog


def _weights():
    pe, inj, total = make_catalog(3, 16, 40, seed=2)
    zm = M.PowerlawRedshiftModel(pe["redshift"], inj["redshift"])

    def w(d):
        return M.powerlaw_primary_ratio_pdf(d["mass_1"], d["mass_ratio"], -2.0, 1.0, 5.0, 100.0) * zm(d["redshift"], 2.7) / d["prior"]

    return w(pe), w(inj), zm, total


def test_max_variance_cut_argument_check_matches_reference():
    pw, iw, zm, total = _weights()
    with pytest.raises(ValueError, match="max_variance_cut is True which requires"):
        hierarchical_likelihood(pw, iw, total, 3, 1.0, surveyed_hypervolume=zm.normalization(2.7), max_variance_cut=True)  # min_neff_cut defaults to True


def test_out_of_scope_branches_raise():
    pw, iw, zm, total = _weights()
    with pytest.raises(NotImplementedError):
        hierarchical_likelihood(pw, iw, total, 3, 1.0, surveyed_hypervolume=zm.normalization(2.7), categorical=True)
    with pytest.raises(TypeError):
        hierarchical_likelihood(np.ones((3, 16)), np.ones(40), total, 3, 1.0, surveyed_hypervolume=zm.normalization(2.7))


def test_density_algebra():
    pw, iw, zm, total = _weights()
    assert isinstance(pw, Density) and pw.side == "pe" and iw.side == "inj"
    assert len(pw.factors) == 3 and len(pw.log_static) == 1  # PL q, PL m1, PL z; / prior
    half = 0.5 * pw
    assert half.log_const == pytest.approx(np.log(0.5))
    with pytest.raises(ValueError):
        pw * iw  # PE and injection products cannot be mixed


def test_model_shapes_and_truncation_like_reference_tests():
    """tests/models/bsplines/separable_test.py:93-97 and parametric_test.py: masks mark samples
    outside [mmin, mmax] / z > zmax as zero-density."""
    pe, inj, _ = make_catalog(3, 16, 40, seed=2)
    m = M.BSplineMass(10, pe["mass_1"], inj["mass_1"], mmin=5.0, mmax=100.0)
    f = m(np.zeros(10), pe_samples=True).factors[0]
    assert f.mask.shape == pe["mass_1"].shape
    assert np.array_equal(f.mask, (pe["mass_1"] >= 5.0) & (pe["mass_1"] <= 100.0))
    zm = M.PowerlawRedshiftModel(pe["redshift"], inj["redshift"])
    fz = zm(pe["redshift"], 2.0).factors[0]
    assert np.array_equal(fz.mask, pe["redshift"] <= zm.zmax)
    with pytest.raises(ValueError):
        zm(pe["redshift"][:, :3], 2.0)  # not the array the model was built with
