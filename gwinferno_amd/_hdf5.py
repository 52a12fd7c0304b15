"""Minimal ctypes binding of the HDF5 C library (libhdf5), enough to READ the NetCDF-4 files arviz writes for an
InferenceData (gwinferno/preprocess/data_collection.py:203-207 -> ``idata.to_netcdf``): groups, n-dimensional numeric
datasets, fixed- and variable-length string datasets, scalar attributes.  h5py / netCDF4 / xarray / arviz are not needed;
the shared library is looked up as ``$GWI_HDF5_LIB``, the loader's ``libhdf5.so*`` and ``/opt/conda/lib``.  The handful
of creation calls the fixture generator needs (tests/golden/make_idata_fixture.py) are bound here too."""
import ctypes as C
import ctypes.util
import glob
import os

import numpy as np

hid_t = C.c_int64  # HDF5 >= 1.10
herr_t = C.c_int
hsize_t = C.c_uint64
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL = 0, 0
H5T_VARIABLE = C.c_size_t(-1).value
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3  # H5T_class_t
H5S_SCALAR = 0

_lib = None


class Hdf5Unavailable(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    cands = [os.environ.get("GWI_HDF5_LIB"), ctypes.util.find_library("hdf5")] + sorted(glob.glob("/opt/conda/lib/libhdf5.so*")) + sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libhdf5*.so*"))
    err = None
    for c in cands:
        if not c:
            continue
        try:
            L = C.CDLL(c)
            L.H5open.restype = herr_t
            if L.H5open() < 0:
                raise OSError("H5open failed")
            break
        except OSError as exc:
            err = exc
    else:
        raise Hdf5Unavailable(f"no usable libhdf5 (set GWI_HDF5_LIB): {err}")
    sig = {
        "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]), "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fclose": (herr_t, [hid_t]),
        "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gclose": (herr_t, [hid_t]),
        "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Dclose": (herr_t, [hid_t]), "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
        "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]), "H5Dvlen_reclaim": (herr_t, [hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]), "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Screate": (hid_t, [C.c_int]), "H5Sclose": (herr_t, [hid_t]),
        "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_size": (C.c_size_t, [hid_t]), "H5Tis_variable_str": (C.c_int, [hid_t]), "H5Tcopy": (hid_t, [hid_t]),
        "H5Tset_size": (herr_t, [hid_t, C.c_size_t]), "H5Tclose": (herr_t, [hid_t]), "H5Tget_sign": (C.c_int, [hid_t]),
        "H5Aexists": (C.c_int, [hid_t, C.c_char_p]), "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Aget_type": (hid_t, [hid_t]), "H5Aget_space": (hid_t, [hid_t]),
        "H5Aread": (herr_t, [hid_t, hid_t, C.c_void_p]), "H5Aclose": (herr_t, [hid_t]), "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]),
        "H5Awrite": (herr_t, [hid_t, hid_t, C.c_void_p]), "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
        "H5Pcreate": (hid_t, [hid_t]), "H5Pset_chunk": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t)]), "H5Pset_deflate": (herr_t, [hid_t, C.c_uint]), "H5Pclose": (herr_t, [hid_t]),
        "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
        "H5Gget_info": (herr_t, [hid_t, C.c_void_p]), "H5Lget_name_by_idx": (C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t]),
        "H5Aget_num_attrs": (C.c_int, [hid_t]), "H5Aopen_by_idx": (hid_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, hid_t, hid_t]),
        "H5Aget_name": (C.c_ssize_t, [hid_t, C.c_size_t, C.c_char_p]), "H5Oopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Oclose": (herr_t, [hid_t]),
        "H5Iget_type": (C.c_int, [hid_t]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    L.H5Eset_auto2(0, None, None)  # errors come back as negative returns; no stack dumps on stderr
    for g in ("H5T_NATIVE_DOUBLE_g", "H5T_NATIVE_FLOAT_g", "H5T_NATIVE_INT64_g", "H5T_NATIVE_INT32_g", "H5T_C_S1_g", "H5P_CLS_DATASET_CREATE_ID_g"):
        setattr(L, g[:-2], hid_t.in_dll(L, g).value)
    _lib = L
    return L


def _check(v, what):
    if v < 0:
        raise OSError(f"HDF5: {what} failed")
    return v


def _shape(L, space):
    nd = _check(L.H5Sget_simple_extent_ndims(space), "H5Sget_simple_extent_ndims")
    dims = (hsize_t * max(nd, 1))()
    if nd:
        L.H5Sget_simple_extent_dims(space, dims, None)
    return tuple(int(d) for d in dims[:nd])


def _read(L, obj, ftype, shape, reader, space=None):
    """Read a dataset / attribute `obj` of file type `ftype` into NumPy (numbers) or a list of str (strings).  `space`
    (the object's dataspace) lets variable-length string reads hand the library's buffers back (H5Dvlen_reclaim)."""
    cls = L.H5Tget_class(ftype)
    n = int(np.prod(shape)) if shape else 1
    if cls == H5T_STRING:
        if L.H5Tis_variable_str(ftype) > 0:
            mtype = L.H5Tcopy(L.H5T_C_S1)
            L.H5Tset_size(mtype, H5T_VARIABLE)
            buf = (C.c_char_p * n)()
            _check(reader(mtype, buf), "read (variable-length strings)")
            out = [b.decode() if b is not None else "" for b in buf]  # copies: the pointers below still belong to the library
            if space is not None:
                L.H5Dvlen_reclaim(mtype, space, H5P_DEFAULT, C.cast(buf, C.c_void_p))
            L.H5Tclose(mtype)
        else:
            size = L.H5Tget_size(ftype)
            raw = C.create_string_buffer(size * n)
            _check(reader(ftype, raw), "read (fixed-length strings)")
            out = [raw.raw[i * size : (i + 1) * size].split(b"\0")[0].decode() for i in range(n)]
        return np.array(out, dtype=object).reshape(shape) if shape else out[0]
    if cls == H5T_FLOAT:
        arr, mtype = np.empty(shape, dtype=np.float64), L.H5T_NATIVE_DOUBLE
    elif cls == H5T_INTEGER:
        arr, mtype = np.empty(shape, dtype=np.int64), L.H5T_NATIVE_INT64
    else:
        raise OSError(f"HDF5: unsupported datatype class {cls}")
    _check(reader(mtype, arr.ctypes.data_as(C.c_void_p)), "read")
    return arr if shape else arr[()]


class File:
    """Read-only view: ``f["group/dataset"]`` -> ndarray / array of str; ``f.attr("group", "name")`` -> scalar."""

    def __init__(self, path):
        self.L = lib()
        self.fid = _check(self.L.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT), f"H5Fopen({path})")
        self.skipped_attributes = []  # attributes attrs() could not decode (unsupported datatype classes)

    def close(self):
        if self.fid:
            self.L.H5Fclose(self.fid)
            self.fid = 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def exists(self, path):
        cur = ""
        for part in path.strip("/").split("/"):
            cur += "/" + part
            if self.L.H5Lexists(self.fid, cur.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    def __getitem__(self, path):
        L = self.L
        d = _check(L.H5Dopen2(self.fid, path.encode(), H5P_DEFAULT), f"H5Dopen2({path})")
        space, ftype = L.H5Dget_space(d), L.H5Dget_type(d)
        try:
            return _read(L, d, ftype, _shape(L, space), lambda mt, buf: L.H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf), space)
        finally:
            L.H5Tclose(ftype), L.H5Sclose(space), L.H5Dclose(d)

    def keys(self, group="/"):
        """Names of the links in a group (H5_INDEX_NAME order)."""
        L = self.L

        class Info(C.Structure):
            _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_int)]

        g = _check(L.H5Gopen2(self.fid, group.encode(), H5P_DEFAULT), f"H5Gopen2({group})")
        try:
            info = Info()
            _check(L.H5Gget_info(g, C.byref(info)), "H5Gget_info")
            out = []
            for i in range(int(info.nlinks)):
                n = _check(L.H5Lget_name_by_idx(g, b".", 0, 0, i, None, 0, H5P_DEFAULT), "H5Lget_name_by_idx")
                buf = C.create_string_buffer(n + 1)
                L.H5Lget_name_by_idx(g, b".", 0, 0, i, buf, n + 1, H5P_DEFAULT)
                out.append(buf.value.decode())
            return out
        finally:
            L.H5Gclose(g)

    def is_dataset(self, path):
        o = self.L.H5Oopen(self.fid, path.encode(), H5P_DEFAULT)
        if o < 0:
            return False
        try:
            return self.L.H5Iget_type(o) == 5  # H5I_DATASET
        finally:
            self.L.H5Oclose(o)

    def attrs(self, path="/"):
        """All attributes of a group / dataset / the file root as a dict."""
        L = self.L
        o = _check(L.H5Oopen(self.fid, path.encode(), H5P_DEFAULT), f"H5Oopen({path})")
        try:
            out = {}
            for i in range(max(L.H5Aget_num_attrs(o), 0)):
                a = _check(L.H5Aopen_by_idx(o, b".", 0, 0, i, H5P_DEFAULT, H5P_DEFAULT), "H5Aopen_by_idx")
                n = L.H5Aget_name(a, 0, None)
                buf = C.create_string_buffer(n + 1)
                L.H5Aget_name(a, n + 1, buf)
                space, ftype = L.H5Aget_space(a), L.H5Aget_type(a)
                try:
                    v = _read(L, a, ftype, _shape(L, space), lambda mt, b_: L.H5Aread(a, mt, b_), space)
                    out[buf.value.decode()] = v[0] if isinstance(v, np.ndarray) and v.size == 1 else v
                except OSError:
                    # a datatype this reader does not cover (references, compounds: netCDF-4's dimension bookkeeping): not in
                    # the dict; attr(path, name) on such an attribute raises with the datatype class
                    self.skipped_attributes.append(f"{path}@{buf.value.decode()}")
                finally:
                    L.H5Tclose(ftype), L.H5Sclose(space), L.H5Aclose(a)
            return out
        finally:
            L.H5Oclose(o)

    def attr(self, group, name, default=None):
        L = self.L
        g = _check(L.H5Gopen2(self.fid, group.encode(), H5P_DEFAULT), f"H5Gopen2({group})")
        try:
            if L.H5Aexists(g, name.encode()) <= 0:
                if default is not None:
                    return default
                raise KeyError(f"attribute {name!r} of {group!r}")
            a = _check(L.H5Aopen(g, name.encode(), H5P_DEFAULT), "H5Aopen")
            space, ftype = L.H5Aget_space(a), L.H5Aget_type(a)
            try:
                v = _read(L, a, ftype, _shape(L, space), lambda mt, buf: L.H5Aread(a, mt, buf), space)
                return v[0] if isinstance(v, np.ndarray) and v.size == 1 else v  # netCDF stores scalars as 1-element arrays
            finally:
                L.H5Tclose(ftype), L.H5Sclose(space), L.H5Aclose(a)
        finally:
            L.H5Gclose(g)
