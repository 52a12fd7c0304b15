"""CPU: catalog ingestion (gwinferno_amd/catalog.py) against the outputs of the UNMODIFIED reference
functions (preprocess/selection.py:12-142, preprocess/data_collection.py:93-98) recorded in
tests/golden/catalog.npz by tests/golden/make_golden.py (in-memory h5py/xarray stand-ins)."""
import os

import numpy as np
import pytest

from gwinferno_amd import catalog as cat

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "catalog.npz"))


def _table(prefix):
    return {k[len(prefix):]: GOLD[k] for k in GOLD.files if k.startswith(prefix)}


@pytest.mark.parametrize("tag,names,kw", [
    ("mq", ["mass_1", "mass_ratio", "redshift"], {}),
    ("spins", ["mass_1", "mass_ratio", "redshift", "a_1", "a_2", "cos_tilt_1", "cos_tilt_2"], {"ifar": 2.0, "snr": 12.0}),
    ("cuts", ["mass_1", "redshift"], {"additional_cuts": {"pastro_cwb": 0.9}}),
])
def test_o3_found_injections_match_reference(tag, names, kw):
    injs, found = cat.o3_found_injections(_table("o3_in/"), names, **kw)
    ref = GOLD[f"o3_out/{tag}/data"]
    params = [str(p) for p in GOLD[f"o3_out/{tag}/params"]]
    assert list(injs) == params
    assert int(found.sum()) == ref.shape[1] and 0 < found.sum() < found.size
    for i, p in enumerate(params):
        assert np.array_equal(injs[p], ref[i]) or np.allclose(injs[p], ref[i], rtol=1e-15, atol=0), p
    total, t_years = GOLD[f"o3_out/{tag}/attrs"]
    assert cat.analysis_time_years({"analysis_time_s": GOLD["o3_attrs"][1]}) == t_years


@pytest.mark.parametrize("tag,names,kw", [("mq", ["mass_1", "mass_ratio", "redshift"], {}), ("spins", ["mass_1", "mass_ratio", "redshift", "a_1"], {"ifar": 0.5, "snr": 11.0})])
def test_o4a_found_injections_match_reference(tag, names, kw):
    injs, found = cat.o4a_found_injections(_table("o4a_in/"), names, **kw)
    ref = GOLD[f"o4a_out/{tag}/data"]
    params = [str(p) for p in GOLD[f"o4a_out/{tag}/params"]]
    assert list(injs) == params
    assert int(found.sum()) == ref.shape[1] and 0 < found.sum() < found.size
    for i, p in enumerate(params):
        assert np.allclose(injs[p], ref[i], rtol=1e-15, atol=0), p


def test_redshift_prior_tables_match_reference():
    z = GOLD["pz/z"]
    assert np.allclose(cat.dl_2_prior_on_z(z), GOLD["pz/comoving"], rtol=1e-12)
    assert np.allclose(cat.dl_2_prior_on_z(z, euclidean=True), GOLD["pz/euclidean"], rtol=1e-12)


def test_pe_sampling_prior_formula():
    """data_collection.py:101-132 spelled out for a two-event tensor with mixed redshift priors."""
    rng = np.random.default_rng(3)
    pe = {"redshift": rng.uniform(0.05, 1.5, (2, 50)), "mass_1": rng.uniform(5, 80, (2, 50))}
    got = cat.pe_sampling_prior(pe, ["mass_1", "mass_ratio", "redshift", "a_1"], redshift_prior=["euclidean", "comoving"])
    zs = np.linspace(0, 1.9 * 1.01, 1000)
    for i, eu in enumerate((True, False)):
        p = cat.dl_2_prior_on_z(zs, euclidean=eu)
        want = np.interp(pe["redshift"][i], zs, p / np.trapezoid(p, zs)) * (1 + pe["redshift"][i]) ** 2 * pe["mass_1"][i] / 4
        assert np.allclose(got[i], want, rtol=1e-14)
    with pytest.raises(AssertionError):
        cat.pe_sampling_prior(pe, ["redshift"], redshift_prior="flat")


def test_read_pe_netcdf3_roundtrip(tmp_path):
    """The reference's PE tensor layout (NetCDF-3 classic: a char `param` coordinate + one (param, sample)
    float32 variable per event), written here with scipy and read back."""
    from scipy.io import netcdf_file

    params = ["mass_1", "mass_ratio", "redshift", "prior"]
    rng = np.random.default_rng(0)
    events = {f"GW{150914 + i}": rng.uniform(0.1, 50.0, (len(params), 12)).astype(">f4") for i in range(3)}
    path = str(tmp_path / "pe.h5")
    with netcdf_file(path, "w") as f:
        f.createDimension("param", len(params))
        f.createDimension("sample", 12)
        f.createDimension("string10", 10)
        v = f.createVariable("param", "S1", ("param", "string10"))
        for i, p in enumerate(params):
            v[i] = np.array(list(p.ljust(10)), dtype="S1")
        f.createVariable("sample", "i4", ("sample",))[:] = np.arange(12)
        for name, arr in events.items():
            f.createVariable(name, ">f4", ("param", "sample"))[:] = arr
    pedict, names = cat.read_pe_netcdf3(path, n_samples=8)
    assert names == list(events) and list(pedict) == params
    for i, p in enumerate(params):
        assert pedict[p].shape == (3, 8) and pedict[p].dtype == np.float64
        assert np.array_equal(pedict[p], np.stack([events[n][i, :8].astype(np.float64) for n in names]))


_REF_PE = "/root/reference/tests/data/xarray_GWTC3_BBH_69evs_downsampled_1000samps_nospin.h5"


@pytest.mark.skipif(not os.path.exists(_REF_PE), reason="reference tree only exists in the build container")
def test_reader_on_the_reference_pe_file():
    """The reference's own GWTC-3 PE tensor, read by the product reader, equals the PE arrays stored in the
    committed GWTC-3 golden case (which the generator read independently)."""
    from golden_util import GoldenCase

    pe, events = cat.read_pe_netcdf3(_REF_PE, n_samples=64)
    case = GoldenCase("gwtc3_pl_test")
    assert len(events) == 69
    for k, v in case.pe.items():
        assert np.array_equal(pe[k], v), k


def _hdf5_or_skip():
    from gwinferno_amd import _hdf5

    try:
        _hdf5.lib()
    except _hdf5.Hdf5Unavailable as exc:
        pytest.skip(str(exc))
    return _hdf5


def test_inference_data_file_loads_like_the_reference_loader():
    """f4: ``load_pe_and_injections_as_dict`` (pipeline/utils.py:51-96) on the InferenceData layout of
    preprocess/data_collection.py:203-207 -- a NetCDF-4 (HDF5) file with groups pe_data / inj_data.  The fixture
    (tests/golden/idata_small.h5) was written by the HDF5 C library itself (tests/golden/make_idata_fixture.py: variable-length
    string coordinates, a chunked + deflated variable, 1-element attribute arrays as netCDF-4 stores them)."""
    _hdf5_or_skip()
    from gwinferno_amd.catalog import load_pe_and_injections_as_dict, read_hdf5_group

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    want = np.load(os.path.join(here, "idata_small.npz"))
    path = os.path.join(here, "idata_small.h5")
    pedict, injdict, constants, names = load_pe_and_injections_as_dict(path)
    assert names == list(want["params"]) and list(pedict) == names and list(injdict) == names
    for i, k in enumerate(names):
        assert pedict[k].shape == (5, 40) and pedict[k].flags.c_contiguous
        assert np.array_equal(pedict[k], want["posteriors"][:, i, :])
        assert np.array_equal(injdict[k], want["injections"][i])
    assert constants == {"total_inj": int(want["total_generated"]), "obs_time": float(want["analysis_time"]), "nObs": 5}
    # ignore=[...] drops events by name (:76-81); nObs still counts the events of the file, as in the reference
    drop = [str(want["events"][1]), str(want["events"][4])]
    pe2, _, c2, _ = load_pe_and_injections_as_dict(path, ignore=drop)
    assert pe2["mass_1"].shape == (3, 40) and np.array_equal(pe2["mass_1"], want["posteriors"][[0, 2, 3], 0, :]) and c2["nObs"] == 5
    # the generic group reader (LVK injection files: selection.py:24-36)
    cols, attrs, _ = read_hdf5_group(path, "inj_data")
    assert set(cols) == {"injections", "param", "injection"} and attrs["total_generated"] == int(want["total_generated"]) and abs(attrs["analysis_time"] - 0.75) < 1e-15
    assert list(cols["param"]) == names and np.array_equal(cols["injection"], np.arange(300))
    # ... and the loaded catalog binds to a model like any other
    from gwinferno_amd.compositions import COMPOSITIONS

    comp = COMPOSITIONS["plpeak"](pedict, injdict)
    assert comp.pe["mass_1"].shape == (5, 40)


@pytest.mark.parametrize("comp_name", ["plpeak_full", "bspline_test"])
def test_inference_data_file_through_the_oracle(comp_name):
    """CPU half of tests/test_gpu_formats.py: the catalog read from tests/golden/idata_small.h5 by the product reader, bound
    by the product binder and evaluated by the C oracle reproduces the sites the unmodified reference computed from the same
    arrays (tests/golden/idata_golden.npz, written by make_golden.py formats)."""
    _hdf5_or_skip()
    from golden_util import GOLDEN_DIR, rel_err

    from gwinferno_amd.catalog import load_pe_and_injections_as_dict
    from gwinferno_amd.compositions import COMPOSITIONS
    from gwinferno_amd.engine import bind
    from oracle.c_oracle import COracle

    gold = np.load(os.path.join(GOLDEN_DIR, "idata_golden.npz"))
    pedict, injdict, constants, _ = load_pe_and_injections_as_dict(os.path.join(GOLDEN_DIR, "idata_small.h5"))
    comp = COMPOSITIONS[comp_name](pedict, injdict)
    pre = f"{comp_name}/theta/"
    thetas = {k[len(pre):]: gold[k] for k in gold.files if k.startswith(pre)}
    for i in range(2):
        p = {k: (v[i] if v.ndim > 1 else float(v[i])) for k, v in thetas.items()}
        bm = bind(comp.weights(p, True), comp.weights(p, False), comp.hypervolume(p))
        got = COracle(bm).evaluate(bm.theta_of(comp.weights(p, True)), constants["total_inj"], nobs=constants["nObs"], min_neff_cut=False)
        assert rel_err(got["log_likelihood"], gold[f"{comp_name}/sites/log_likelihood"][i]) < 1e-9
        assert rel_err(got["logBFs"], gold[f"{comp_name}/sites/logBFs"][i]) < 1e-9
        assert rel_err(got["log_nEffs"], gold[f"{comp_name}/sites/log_nEffs"][i]) < 1e-9


def test_truncated_reference_pe_tensor_equals_the_golden_arrays():
    """tests/golden/gwtc3_first64.nc (the reference's own GWTC-3 PE tensor, first 64 samples per event, NetCDF-3) read by the
    product reader == the PE arrays of the committed GWTC-3 golden case."""
    from golden_util import GOLDEN_DIR, GoldenCase

    pe, events = cat.read_pe_netcdf3(os.path.join(GOLDEN_DIR, "gwtc3_first64.nc"))
    case = GoldenCase("gwtc3_pl_test")
    assert len(events) == 69
    for k, v in case.pe.items():
        assert np.array_equal(pe[k], v), k
