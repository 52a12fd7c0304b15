"""`jax.numpy` stand-in: the NumPy namespace, with array results re-viewed as an ndarray
subclass that offers JAX's functional `.at[idx].set(v)` update and returns NumPy scalars for
0-d results (so `round()`, `float()` ... behave as they do on jax scalars)."""
import numpy as _np


class _AtIndexer:
    __slots__ = ("arr", "idx")

    def __init__(self, arr, idx=None):
        self.arr, self.idx = arr, idx

    def __getitem__(self, idx):
        return _AtIndexer(self.arr, idx)

    def set(self, value):
        out = _np.array(self.arr, copy=True)
        out[self.idx] = value
        return out.view(Arr)

    def add(self, value):
        out = _np.array(self.arr, copy=True)
        _np.add.at(out, self.idx, value)
        return out.view(Arr)

    def get(self):
        return _wrap(_np.asarray(self.arr)[self.idx])


class Arr(_np.ndarray):
    @property
    def at(self):
        return _AtIndexer(self)

    def __getitem__(self, idx):
        # JAX clamps out-of-bounds integer indices on reads (the reference's cumtrapz,
        # numpyro_distributions.py:20-24, relies on it: it reads y[len(y)])
        if isinstance(idx, (int, _np.integer)) and self.ndim >= 1 and not isinstance(idx, bool):
            n = self.shape[0]
            if idx >= n:
                idx = n - 1
            elif idx < -n:
                idx = 0
        return _np.ndarray.__getitem__(self, idx)

    def __iter__(self):  # ndarray iterates through the sequence protocol, which the clamp above would never end
        if self.ndim == 0:
            raise TypeError("iteration over a 0-d array")
        return (_np.ndarray.__getitem__(self, i) for i in range(self.shape[0]))

    def __array_wrap__(self, obj, context=None, return_scalar=False):
        if obj.ndim == 0:
            return obj[()]
        return _np.ndarray.__array_wrap__(self, obj, context, return_scalar)


def _wrap(x):
    if isinstance(x, _np.ndarray):
        if x.ndim == 0:
            return x[()]
        return x.view(Arr)
    if isinstance(x, tuple):
        return tuple(_wrap(v) for v in x)
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def _lift(fn):
    def inner(*a, **k):
        return _wrap(fn(*a, **k))

    inner.__name__ = getattr(fn, "__name__", "fn")
    return inner


_PASSTHROUGH = {"ndarray", "dtype", "float64", "float32", "int32", "int64", "bool_", "newaxis", "pi", "inf", "nan", "e", "linalg"}


def __getattr__(name):
    obj = getattr(_np, name)
    if name in _PASSTHROUGH or not callable(obj) or isinstance(obj, type):
        return obj
    return _lift(obj)


pi = _np.pi
inf = _np.inf
nan = _np.nan
e = _np.e
newaxis = _np.newaxis
ndarray = _np.ndarray
float64 = _np.float64
float32 = _np.float32
int32 = _np.int32
int64 = _np.int64
bool_ = _np.bool_


class _Linalg:
    def __getattr__(self, name):
        return _lift(getattr(_np.linalg, name))


linalg = _Linalg()


def array(obj, dtype=None, **k):
    return _wrap(_np.array(obj, dtype=dtype))


def asarray(obj, dtype=None, **k):
    return _wrap(_np.asarray(obj, dtype=dtype))


def trapezoid(y, x=None, dx=1.0, axis=-1):
    return _wrap(_np.trapezoid(y, x=x, dx=dx, axis=axis))


def where(cond, x=None, y=None):
    if x is None and y is None:
        return _np.where(cond)
    with _np.errstate(all="ignore"):
        return _wrap(_np.where(cond, x, y))
