"""Catalog ingestion for the engine: the data formats and selection logic that
sit in front of the likelihood, restated with NumPy/SciPy only so that real catalogs can feed the engine
where xarray / arviz / h5py are not installed.

* :func:`read_pe_netcdf3` -- the reference's downsampled PE tensor file
  (``tests/data/xarray_GWTC3_BBH_69evs_downsampled_1000samps_nospin.h5``: despite the suffix a NetCDF-3
  classic file written by xarray; one ``(param, sample)`` float32 variable per event plus a character
  ``param`` coordinate) -> ``pedict[param] : (N_ev, N_pe)`` as ``pipeline/utils.py:82-96`` builds it.
* :func:`o3_found_injections` / :func:`o4a_found_injections` -- the found-injection cuts and the sampling
  prior of ``preprocess/selection.py:82-142`` / ``:12-80``, applied to a plain mapping ``name -> array``
  (the columns of the HDF5 ``injections`` group, or the structured ``events`` table): returns ``injdict``
  with the keys the likelihood expects (``mass_1, mass_2, mass_ratio, redshift[, a_i, cos_tilt_i], prior``).
* :func:`dl_2_prior_on_z`, :func:`pe_sampling_prior` -- the per-sample PE prior of
  ``preprocess/data_collection.py:93-132``.

* :func:`load_pe_and_injections_as_dict` -- the reference's loader of the same name (``pipeline/utils.py:51-96``) for the
  arviz InferenceData file (NetCDF-4 = HDF5, groups ``pe_data`` / ``inj_data``) that
  ``preprocess/data_collection.py:203-207`` writes: same arguments, same four return values, read through the HDF5 C library
  itself (:mod:`gwinferno_amd._hdf5`, ctypes) -- no h5py, xarray or arviz.

Reading the LVK HDF5 injection files goes through the same binding (:func:`read_hdf5_group`, h5py if it happens to be
installed); the cut and prior arithmetic needs neither.
"""
import numpy as np

from .cosmology import planck15_lvk

def read_pe_netcdf3(path, n_samples=None):
    """``pedict[param] -> (N_ev, N_pe)`` float64 from the reference's NetCDF-3 PE tensor; also returns the
    event names.  ``n_samples`` keeps the first samples of every event."""
    from scipy.io import netcdf_file

    with netcdf_file(path, "r", mmap=False) as f:
        names = [b"".join(row).decode().strip() for row in f.variables["param"].data]
        events = [k for k in f.variables if k not in ("param", "sample", "samples")]
        data = np.stack([np.asarray(f.variables[e].data, dtype=np.float64)[:, :n_samples] for e in events])
    return {n: np.ascontiguousarray(data[:, i, :]) for i, n in enumerate(names)}, events


def read_hdf5_group(path, group):
    """Columns and attributes of one HDF5 group as plain dicts -- ``(columns, group attributes, file attributes)``, what
    ``preprocess/selection.py:24-36, :105-118`` reads from the LVK injection files -- through the HDF5 C library
    (:mod:`gwinferno_amd._hdf5`)."""
    from . import _hdf5

    with _hdf5.File(path) as f:
        cols = {k: f[f"{group}/{k}"] for k in f.keys(group) if f.is_dataset(f"{group}/{k}")}
        return cols, f.attrs(group), f.attrs("/")


def load_pe_and_injections_as_dict(file, ignore=None):
    """``pipeline/utils.py:51-96``: ``(pedict, injdict, constants, param_names)`` from the InferenceData file
    ``save_posterior_samples_and_injection_datasets_as_idata`` wrote.  ``pedict[param] : (N_ev, N_pe)`` =
    ``pe_data.posteriors.sel(param=k).values`` (events named in ``ignore`` dropped, :76-81), ``injdict[param] : (N_inj,)`` =
    ``inj_data.injections.sel(param=k).values``, ``constants = {total_inj, obs_time, nObs}`` from the ``inj_data``
    attributes ``total_generated`` / ``analysis_time`` and the number of events (:88-92; as in the reference ``nObs``
    counts the events in the file, ignored ones included)."""
    from . import _hdf5

    with _hdf5.File(file) as f:
        post = np.asarray(f["pe_data/posteriors"], dtype=np.float64)  # (event, param, samples)
        pe_params = [str(p) for p in f["pe_data/param"]]
        events = np.array([str(e) for e in f["pe_data/event"]]) if f.exists("pe_data/event") else np.arange(post.shape[0]).astype(str)
        injs = np.asarray(f["inj_data/injections"], dtype=np.float64)  # (param, injection)
        inj_params = [str(p) for p in f["inj_data/param"]]
        total_inj = f.attr("inj_data", "total_generated")
        obs_time = f.attr("inj_data", "analysis_time")
    if post.ndim != 3 or post.shape[1] != len(pe_params) or injs.ndim != 2 or injs.shape[0] != len(inj_params):
        raise ValueError(f"{file}: posteriors {post.shape} / injections {injs.shape} do not match their `param` coordinates")
    sel = np.ones(post.shape[0], dtype=bool)
    if ignore is not None:
        for gw in ignore:
            sel &= events != gw
    pedict = {k: np.ascontiguousarray(post[sel, i, :]) for i, k in enumerate(pe_params)}
    injdict = {k: np.ascontiguousarray(injs[i]) for i, k in enumerate(inj_params)}
    constants = {"total_inj": total_inj, "obs_time": obs_time, "nObs": post.shape[0]}
    return pedict, injdict, constants, list(pe_params)


def _spins(table, found, injs, param_names, zeros_ok):
    """a_i, cos_tilt_i and the isotropic-spin factor of the prior (selection.py:62-68, 119-128)."""
    if ("a_1" in param_names) | ("chi_eff" in param_names):
        n_found = int(np.sum(found))
        for ii in (1, 2):
            comp = []
            for ax in "xyz":
                key = f"spin{ii}{ax}"
                if zeros_ok and ax != "z" and key not in table:
                    comp.append(np.zeros(n_found))  # aligned-spin injection sets carry no x/y columns (:121-123)
                else:
                    comp.append(np.asarray(table[key])[found])
            injs[f"a_{ii}"] = (comp[0] ** 2 + comp[1] ** 2 + comp[2] ** 2) ** 0.5
            injs[f"cos_tilt_{ii}"] = comp[2] / injs[f"a_{ii}"]
        injs["prior"] = injs["prior"] * (2 * np.pi * injs["a_1"] ** 2) * (2 * np.pi * injs["a_2"] ** 2)


def o3_found_injections(table, param_names, ifar=1, snr=10, additional_cuts=None):
    """selection.py:82-142 on the columns of the ``injections`` group.  An injection is found if ANY
    pipeline's ``ifar*`` exceeds ``ifar``, or it is an O1/O2 injection (``name``) with
    ``optimal_snr_net > snr``, or any ``additional_cuts[key]`` threshold is met (``>=``).  Returns
    ``(injdict, found_mask)``; the prior is ``sampling_pdf`` times the isotropic-spin Jacobian (when spin
    magnitudes are used) times ``m1`` (when working in mass ratio)."""
    m1 = np.asarray(table["mass1_source"])
    found = np.zeros_like(m1, dtype=bool)
    for key in table:
        if "ifar" in key.lower():
            found = found | (np.asarray(table[key]) > ifar)
        if "name" in table:
            name = np.asarray(table["name"])
            gwtc1 = (name == b"o1") | (name == b"o2")
            found = found | (gwtc1 & (np.asarray(table["optimal_snr_net"]) > snr))
    if additional_cuts is not None:
        for k in additional_cuts:
            found = found | (np.asarray(table[k]) >= additional_cuts[k])
    m2 = np.asarray(table["mass2_source"])
    injs = dict(mass_1=m1[found], mass_2=m2[found], mass_ratio=m2[found] / m1[found], redshift=np.asarray(table["redshift"])[found])
    injs["prior"] = np.asarray(table["sampling_pdf"])[found]
    _spins(table, found, injs, param_names, zeros_ok=True)
    if "mass_ratio" in param_names:
        injs["prior"] = injs["prior"] * m1[found]
    return injs, found


_O4A_LNPDRAW = "lnpdraw_mass1_source_mass2_source_redshift_spin1x_spin1y_spin1z_spin2x_spin2y_spin2z"


def o4a_found_injections(events, param_names, ifar=1, snr=10):
    """selection.py:12-80 on the structured ``events`` table (or a mapping of its fields): found if the
    semianalytic SNR ``>= snr`` or any ``*far*`` field ``<= 1/ifar``; prior ``exp(lnpdraw)/weights`` with the
    same Jacobians as above."""
    names = events.dtype.names if hasattr(events, "dtype") and events.dtype.names else list(events)
    found = np.asarray(events["semianalytic_observed_phase_maximized_snr_net"]) >= snr
    for key in names:
        if "far" in key:
            found = found | (np.asarray(events[key]) <= 1 / ifar)
    m1, m2 = np.asarray(events["mass1_source"]), np.asarray(events["mass2_source"])
    injs = dict(mass_1=m1[found], mass_2=m2[found], mass_ratio=m2[found] / m1[found], redshift=np.asarray(events["redshift"])[found])
    injs["prior"] = np.exp(np.asarray(events[_O4A_LNPDRAW])[found]) / np.asarray(events["weights"])[found]
    if "mass_ratio" in param_names:
        injs["prior"] = injs["prior"] * m1[found]
    _spins(events, found, injs, param_names, zeros_ok=False)
    return injs, found


def analysis_time_years(attrs):
    """selection.py:32-37, 107-112: the observing time in years from whichever attribute carries it (seconds)."""
    for key in ("analysis_time", "total_analysis_time", "analysis_time_s"):
        if key in attrs:
            return float(np.asarray(attrs[key])) / 365.25 / 24 / 60 / 60  # the reference's division order
    raise KeyError("analysis time not found")


def dl_2_prior_on_z(z, euclidean=False):
    """data_collection.py:93-98: the redshift prior implied by a ``d_L^2`` distance prior (Euclidean) or the
    uniform-in-comoving-volume-and-source-time prior ``dVc/dz / (1+z)``."""
    cosmo = planck15_lvk()
    z = np.asarray(z, dtype=np.float64)
    if euclidean:
        dl = cosmo.z_to_DL(z)
        return dl**2 * (dl / (1 + z) + (1 + z) * cosmo.dDc_dz(z))
    return cosmo.dVc_dz(z) / (1 + z)


def pe_sampling_prior(pedict, param_names, redshift_prior="comoving"):
    """data_collection.py:101-132: per-sample PE prior ``(N_ev, N_pe)`` for the parameters in use.
    ``redshift_prior`` is one string or one per event (``"euclidean"`` | ``"comoving"``)."""
    z = np.asarray(pedict["redshift"], dtype=np.float64)
    n_ev = z.shape[0]
    kinds = [redshift_prior] * n_ev if isinstance(redshift_prior, str) else list(redshift_prior)
    prior = np.ones_like(z)
    if "redshift" in param_names:
        z_max = max(1.9, float(np.max(z)))
        zs = np.linspace(0, z_max * 1.01, 1000)
        tables = {}
        for kind in set(kinds):
            if kind not in ("euclidean", "comoving"):
                raise AssertionError("redshift prior not valid. check spelling")
            p = dl_2_prior_on_z(zs, euclidean=kind == "euclidean")
            tables[kind] = p / np.trapezoid(p, zs)
        for i, kind in enumerate(kinds):
            prior[i] *= np.interp(z[i], zs, tables[kind])
    if "mass_1" in param_names:
        prior *= (1 + z) ** 2  # flat in detector-frame component masses
    if "mass_ratio" in param_names:
        prior *= np.asarray(pedict["mass_1"], dtype=np.float64)
    if "a_1" in param_names:
        prior *= 1 / 4
    return prior
