"""GPU: catalog FILES -> product readers -> engine, against what the unmodified reference computes from the same data
(VERDICT r2 item 6).

* ``tests/golden/idata_small.h5`` -- the InferenceData (NetCDF-4 = HDF5) layout ``load_pe_and_injections_as_dict`` reads
  (gwinferno/pipeline/utils.py:51-96), through ``gwinferno_amd.catalog`` / ``_hdf5``; golden sites in
  ``idata_golden.npz`` (tests/golden/make_golden.py formats: the reference's hierarchical_likelihood on the arrays the
  file was written from, with the file's own total_generated).
* ``tests/golden/gwtc3_first64.nc`` -- the reference's GWTC-3 PE tensor (NetCDF-3), first 64 samples per event; golden
  sites are those of ``case_gwtc3_pl_test.npz`` (the generator read the reference's file independently).
"""
import os

import numpy as np
import pytest
from golden_util import GOLDEN_DIR, GoldenCase, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("comp_name", ["plpeak_full", "bspline_test"])
def test_inference_data_file_through_the_engine(comp_name):
    from gwinferno_amd import _hdf5
    from gwinferno_amd.catalog import load_pe_and_injections_as_dict
    from gwinferno_amd.compositions import COMPOSITIONS

    try:
        _hdf5.lib()
    except _hdf5.Hdf5Unavailable as exc:
        pytest.skip(str(exc))
    gold = np.load(os.path.join(GOLDEN_DIR, "idata_golden.npz"))
    pedict, injdict, constants, names = load_pe_and_injections_as_dict(os.path.join(GOLDEN_DIR, "idata_small.h5"))
    assert constants["nObs"] == 5 and pedict["mass_1"].shape == (5, 40) and injdict["mass_1"].shape == (300,)
    comp = COMPOSITIONS[comp_name](pedict, injdict)
    eng = comp.engine()
    pre = f"{comp_name}/theta/"
    thetas = {k[len(pre):]: gold[k] for k in gold.files if k.startswith(pre)}
    tobs = float(gold["tobs_used_by_generator"])
    for i in range(2):
        p = {k: (v[i] if v.ndim > 1 else float(v[i])) for k, v in thetas.items()}
        res = eng.evaluate(comp.theta(p), constants["total_inj"], nobs=constants["nObs"], min_neff_cut=False)
        s = res.summary
        got = {"log_likelihood": s.log_likelihood, "log_l": s.log_l, "logBFs": res.log_bfs, "log_nEffs": res.log_neffs, "log_nEff_inj": s.log_nEff_inj,
               "detection_efficiency": np.exp(s.log_det_eff), "surveyed_hypervolume": s.surveyed_hypervolume_norm / 1e9 * tobs}
        for site, val in got.items():
            assert rel_err(val, gold[f"{comp_name}/sites/{site}"][i]) < 1e-9, (comp_name, i, site)
    eng.close()


def test_reference_netcdf3_pe_tensor_through_the_engine():
    from gwinferno_amd.catalog import read_pe_netcdf3
    from gwinferno_amd.compositions import COMPOSITIONS

    case = GoldenCase("gwtc3_pl_test")
    pe, events = read_pe_netcdf3(os.path.join(GOLDEN_DIR, "gwtc3_first64.nc"))
    assert len(events) == 69 and events[0] == "GW150914"
    for k, v in case.pe.items():  # the reader reproduces the arrays the golden generator read from the reference's file
        assert np.array_equal(pe[k], v), k
    comp = COMPOSITIONS[case.composition](pe, case.inj, mmin=case.meta["mmin"], mmax=case.meta["mmax"])
    eng = comp.engine()
    for i in range(case.n_points):
        res = eng.evaluate(comp.theta(case.point(i)), case.total_inj, min_neff_cut=False)
        s = res.summary
        ref = case.sites["lin"]
        assert rel_err(s.log_likelihood, ref["log_likelihood"][i]) < 1e-9
        assert rel_err(res.log_bfs, ref["logBFs"][i]) < 1e-9 and rel_err(res.log_neffs, ref["log_nEffs"][i]) < 1e-9
        assert rel_err(np.exp(s.log_det_eff), ref["detection_efficiency"][i]) < 1e-9
    eng.close()
