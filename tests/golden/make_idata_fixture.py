#!/usr/bin/env python3
"""Writes tests/golden/idata_small.h5 + idata_small.npz: a small file in the layout
``gwinferno.preprocess.data_collection.save_posterior_samples_and_injection_datasets_as_idata`` produces through arviz
(``az.InferenceData(pe_data=..., inj_data=...).to_netcdf`` -- NetCDF-4 = HDF5): group ``pe_data`` with variable
``posteriors`` (event, param, samples) and string coordinates ``event`` / ``param``; group ``inj_data`` with variable
``injections`` (param, injection), string coordinate ``param`` and the group attributes ``total_generated`` /
``analysis_time`` (``to_dataset(promote_attrs=True)``, data_collection.py:162-200, selection.py:72-76).
Written with the HDF5 C library itself (ctypes; h5py / netCDF4 / arviz do not exist in this image): variable-length
string coordinates and 1-element attribute arrays as netCDF-4 stores them, one variable chunked + deflated.
Run:  python tests/golden/make_idata_fixture.py"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from gwinferno_amd import _hdf5 as H  # noqa: E402
from gwinferno_amd.synthetic import make_catalog  # noqa: E402

PARAMS = ["mass_1", "mass_ratio", "redshift", "a_1", "a_2", "cos_tilt_1", "cos_tilt_2", "prior"]


def write_array(L, loc, name, arr, chunks=None):
    arr = np.ascontiguousarray(arr)
    dims = (H.hsize_t * arr.ndim)(*arr.shape)
    space = L.H5Screate_simple(arr.ndim, dims, None)
    ftype = L.H5T_NATIVE_DOUBLE if arr.dtype == np.float64 else L.H5T_NATIVE_INT64
    dcpl = H.H5P_DEFAULT
    if chunks:
        dcpl = L.H5Pcreate(L.H5P_CLS_DATASET_CREATE_ID)
        L.H5Pset_chunk(dcpl, arr.ndim, (H.hsize_t * arr.ndim)(*chunks))
        L.H5Pset_deflate(dcpl, 4)
    d = L.H5Dcreate2(loc, name.encode(), ftype, space, H.H5P_DEFAULT, dcpl, H.H5P_DEFAULT)
    assert d >= 0
    assert L.H5Dwrite(d, ftype, H.H5S_ALL, H.H5S_ALL, H.H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p)) >= 0
    L.H5Dclose(d), L.H5Sclose(space)
    if chunks:
        L.H5Pclose(dcpl)


def write_strings(L, loc, name, strings):
    vt = L.H5Tcopy(L.H5T_C_S1)
    L.H5Tset_size(vt, H.H5T_VARIABLE)
    dims = (H.hsize_t * 1)(len(strings))
    space = L.H5Screate_simple(1, dims, None)
    d = L.H5Dcreate2(loc, name.encode(), vt, space, H.H5P_DEFAULT, H.H5P_DEFAULT, H.H5P_DEFAULT)
    buf = (C.c_char_p * len(strings))(*[s.encode() for s in strings])
    assert L.H5Dwrite(d, vt, H.H5S_ALL, H.H5S_ALL, H.H5P_DEFAULT, buf) >= 0
    L.H5Dclose(d), L.H5Sclose(space), L.H5Tclose(vt)


def write_attr(L, loc, name, value):
    arr = np.atleast_1d(np.asarray(value))
    space = L.H5Screate_simple(1, (H.hsize_t * 1)(1), None)
    ftype = L.H5T_NATIVE_DOUBLE if arr.dtype.kind == "f" else L.H5T_NATIVE_INT64
    arr = arr.astype(np.float64 if arr.dtype.kind == "f" else np.int64)
    a = L.H5Acreate2(loc, name.encode(), ftype, space, H.H5P_DEFAULT, H.H5P_DEFAULT)
    assert L.H5Awrite(a, ftype, arr.ctypes.data_as(C.c_void_p)) >= 0
    L.H5Aclose(a), L.H5Sclose(space)


def main():
    L = H.lib()
    pe, inj, total = make_catalog(5, 40, 300, seed=2024)
    events = [f"GW1909{15 + i:02d}_{100000 + 7 * i}" for i in range(5)]
    post = np.stack([pe[p] for p in PARAMS], axis=1)            # (event, param, samples)
    injs = np.stack([inj[p] for p in PARAMS], axis=0)           # (param, injection)
    path = os.path.join(HERE, "idata_small.h5")
    f = L.H5Fcreate(path.encode(), H.H5F_ACC_TRUNC, H.H5P_DEFAULT, H.H5P_DEFAULT)
    g = L.H5Gcreate2(f, b"pe_data", H.H5P_DEFAULT, H.H5P_DEFAULT, H.H5P_DEFAULT)
    write_array(L, g, "posteriors", post, chunks=(2, 4, 40))
    write_strings(L, g, "event", events)
    write_strings(L, g, "param", PARAMS)
    write_array(L, g, "samples", np.arange(post.shape[2], dtype=np.int64))
    L.H5Gclose(g)
    g = L.H5Gcreate2(f, b"inj_data", H.H5P_DEFAULT, H.H5P_DEFAULT, H.H5P_DEFAULT)
    write_array(L, g, "injections", injs)
    write_strings(L, g, "param", PARAMS)
    write_array(L, g, "injection", np.arange(injs.shape[1], dtype=np.int64))
    write_attr(L, g, "total_generated", int(total))
    write_attr(L, g, "analysis_time", 0.75)
    L.H5Gclose(g)
    L.H5Fclose(f)
    np.savez_compressed(os.path.join(HERE, "idata_small.npz"), posteriors=post, injections=injs, events=np.array(events), params=np.array(PARAMS), total_generated=int(total),
                        analysis_time=0.75)
    print(f"wrote {path}: {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
