"""jax.numpy of the test shim: NumPy underneath, values wrapped in an opaque Tracer (NOT an ndarray, as real tracers)."""
import numpy as _np

float64, int64, int32 = _np.float64, _np.int64, _np.int32


class Tracer:
    __array_priority__ = 1000

    def __init__(self, val):
        self.val = _np.asarray(val)

    shape = property(lambda self: self.val.shape)
    dtype = property(lambda self: self.val.dtype)
    ndim = property(lambda self: self.val.ndim)

    def __array__(self, dtype=None, copy=None):
        raise TypeError("a traced value was converted to a NumPy array outside pure_callback")

    def __getitem__(self, idx):
        idx = tuple(_unwrap(i) for i in idx) if isinstance(idx, tuple) else _unwrap(idx)
        return Tracer(self.val[idx])

    def __len__(self):
        return len(self.val)

    def _bin(self, other, op):
        return Tracer(op(self.val, _unwrap(other)))

    def __add__(self, o): return self._bin(o, _np.add)  # noqa: E704
    def __radd__(self, o): return self._bin(o, lambda a, b: b + a)  # noqa: E704
    def __sub__(self, o): return self._bin(o, _np.subtract)  # noqa: E704
    def __rsub__(self, o): return self._bin(o, lambda a, b: b - a)  # noqa: E704
    def __mul__(self, o): return self._bin(o, _np.multiply)  # noqa: E704
    def __rmul__(self, o): return self._bin(o, lambda a, b: b * a)  # noqa: E704
    def __truediv__(self, o): return self._bin(o, _np.divide)  # noqa: E704
    def __rtruediv__(self, o): return self._bin(o, lambda a, b: b / a)  # noqa: E704
    def __neg__(self): return Tracer(-self.val)  # noqa: E704


def _unwrap(x):
    return x.val if isinstance(x, Tracer) else x


def _wrap(x):
    return Tracer(x)


def asarray(x, dtype=None):
    return Tracer(_np.asarray(_unwrap(x), dtype=dtype))


array = asarray


def ravel(x):
    return Tracer(_np.ravel(_unwrap(x)))


def concatenate(xs):
    return Tracer(_np.concatenate([_np.atleast_1d(_unwrap(x)) for x in xs]))


def exp(x):
    return Tracer(_np.exp(_unwrap(x)))


def log(x):
    return Tracer(_np.log(_unwrap(x)))


def zeros(shape, dtype=float64):
    return Tracer(_np.zeros(shape, dtype=dtype))
