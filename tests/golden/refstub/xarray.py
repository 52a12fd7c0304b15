"""Stand-in for ``xarray.DataArray`` as the reference's injection loaders construct it
(gwinferno/preprocess/selection.py:71-77, 136-142): keeps data, dims, coords and attrs.  Build container only."""
import numpy as np


class DataArray:
    def __init__(self, data=None, dims=None, coords=None, attrs=None, **kw):
        self.data = np.asarray(data)
        self.values = self.data
        self.dims, self.coords, self.attrs = dims, coords, dict(attrs or {})


class Dataset:
    def __init__(self, *a, **kw):
        raise NotImplementedError("xarray.Dataset is not emulated")


def concat(*a, **kw):
    raise NotImplementedError("xarray.concat is not emulated")
