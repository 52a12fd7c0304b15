"""In-memory stand-in for the few h5py calls of the reference's preprocess modules
(gwinferno/preprocess/selection.py:27-38, 88-135): ``File(path)`` looks ``path`` up in ``REGISTRY``,
which the golden-vector generator fills with ``{"attrs": {...}, "groups": {name: {"attrs": {...},
"data": {key: array}}}, "datasets": {name: array}}``.  Build container only."""
import numpy as np

REGISTRY = {}


class _Attrs(dict):
    def __getitem__(self, k):
        return np.asarray(dict.__getitem__(self, k))


class _Dataset:
    def __init__(self, arr):
        self._a = np.asarray(arr)

    def __getitem__(self, idx):
        return self._a[idx]


class _Group:
    def __init__(self, spec):
        self.attrs = _Attrs(spec.get("attrs", {}))
        self._d = spec["data"]

    def __iter__(self):
        return iter(self._d)

    def keys(self):
        return self._d.keys()

    def __getitem__(self, k):
        return _Dataset(self._d[k])

    def get(self, k, default=None):
        return _Dataset(self._d[k]) if k in self._d else default


class File:
    def __init__(self, path, mode="r"):
        self._spec = REGISTRY[path]
        self.attrs = _Attrs(self._spec.get("attrs", {}))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __getitem__(self, k):
        if k in self._spec.get("groups", {}):
            return _Group(self._spec["groups"][k])
        return _Dataset(self._spec["datasets"][k])
