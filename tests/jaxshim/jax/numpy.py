"""jax.numpy of the test shim: NumPy underneath, values wrapped in an opaque Tracer (NOT an ndarray, as real tracers).

A Tracer records how it was made (parents + the pull-back of each), so that ``jax.value_and_grad`` of the shim can run a
reverse sweep over the handful of operations the adapter and a hand-written potential use; it may carry a leading BATCH
axis (``jax.vmap`` of the shim), which every operation here keeps in front.  float64 / int64 requests are canonicalised to
float32 / int32 while ``jax.config.jax_enable_x64`` is off, as JAX does."""
import numpy as _np

float64, float32, int64, int32 = _np.float64, _np.float32, _np.int64, _np.int32

X64 = [True]  # jax.config.jax_enable_x64 of the shim (tests flip it through jax.config.update)


def canonicalize_dtype(dtype):
    dt = _np.dtype(dtype)
    if not X64[0]:
        if dt == _np.float64:
            return _np.dtype(_np.float32)
        if dt == _np.int64:
            return _np.dtype(_np.int32)
    return dt


class Tracer:
    __array_priority__ = 1000

    def __init__(self, val, parents=(), batched=False):
        self.val = _np.asarray(val)
        self.parents = list(parents)  # [(Tracer, pull-back: cotangent of self -> cotangent of that parent)]
        self.batched = batched        # leading axis = vmap's batch axis

    shape = property(lambda self: self.val.shape[1:] if self.batched else self.val.shape)
    dtype = property(lambda self: self.val.dtype)
    ndim = property(lambda self: self.val.ndim - (1 if self.batched else 0))
    size = property(lambda self: int(_np.prod(self.shape, dtype=_np.int64)))  # (np.size() reads it, as it does on a JAX tracer)

    def __array__(self, dtype=None, copy=None):
        raise TypeError("a traced value was converted to a NumPy array outside pure_callback")

    def __getitem__(self, idx):
        idx = tuple(_unwrap(i) for i in idx) if isinstance(idx, tuple) else _unwrap(idx)
        full = ((slice(None),) + (idx if isinstance(idx, tuple) else (idx,))) if self.batched else idx

        def back(ct, shape=self.val.shape, dtype=self.val.dtype):
            z = _np.zeros(shape, dtype=dtype)
            z[full] = ct
            return z

        return Tracer(self.val[full], [(self, back)], self.batched)

    def __len__(self):
        return self.shape[0]

    def __add__(self, o): return _binary(self, o, _np.add, lambda ct, a, b: ct, lambda ct, a, b: ct)  # noqa: E704
    def __radd__(self, o): return _binary(o, self, _np.add, lambda ct, a, b: ct, lambda ct, a, b: ct)  # noqa: E704
    def __sub__(self, o): return _binary(self, o, _np.subtract, lambda ct, a, b: ct, lambda ct, a, b: -ct)  # noqa: E704
    def __rsub__(self, o): return _binary(o, self, _np.subtract, lambda ct, a, b: ct, lambda ct, a, b: -ct)  # noqa: E704
    def __mul__(self, o): return _binary(self, o, _np.multiply, lambda ct, a, b: ct * b, lambda ct, a, b: ct * a)  # noqa: E704
    def __rmul__(self, o): return _binary(o, self, _np.multiply, lambda ct, a, b: ct * b, lambda ct, a, b: ct * a)  # noqa: E704
    def __truediv__(self, o): return _binary(self, o, _np.divide, lambda ct, a, b: ct / b, lambda ct, a, b: -ct * a / (b * b))  # noqa: E704
    def __rtruediv__(self, o): return _binary(o, self, _np.divide, lambda ct, a, b: ct / b, lambda ct, a, b: -ct * a / (b * b))  # noqa: E704
    def __neg__(self): return Tracer(-self.val, [(self, lambda ct: -ct)], self.batched)  # noqa: E704


def _unwrap(x):
    return x.val if isinstance(x, Tracer) else x


def _wrap(x, batched=False):
    return Tracer(x, batched=batched)


def _unbroadcast(ct, shape):
    """Sum a cotangent back onto an operand of ``shape`` that NumPy broadcasting had stretched."""
    ct = _np.asarray(ct)
    while ct.ndim > len(shape):
        ct = ct.sum(axis=0)
    for ax, n in enumerate(shape):
        if n == 1 and ct.shape[ax] != 1:
            ct = ct.sum(axis=ax, keepdims=True)
    return ct.reshape(shape)


def _binary(a, b, op, back_a, back_b):
    ta, tb = isinstance(a, Tracer), isinstance(b, Tracer)

    def const(x):  # an untraced operand: what JAX would make of it (float64 / int64 canonicalised while x64 is off)
        v = _np.asarray(x)
        return v.astype(canonicalize_dtype(v.dtype)) if v.dtype in (_np.float64, _np.int64) else v

    va, vb = (a.val if ta else const(a)), (b.val if tb else const(b))
    ba, bb = ta and a.batched, tb and b.batched
    batched = ba or bb
    if batched:  # keep the batch axis in front: a batched operand of lower per-example rank gets unit axes behind it
        ra, rb = va.ndim - (1 if ba else 0), vb.ndim - (1 if bb else 0)
        r = max(ra, rb)
        if ba:
            va = va.reshape(va.shape[:1] + (1,) * (r - ra) + va.shape[1:])
        if bb:
            vb = vb.reshape(vb.shape[:1] + (1,) * (r - rb) + vb.shape[1:])
    out = op(va, vb)
    parents = []
    if ta:
        parents.append((a, lambda ct, va=va, vb=vb, shape=a.val.shape: _unbroadcast(back_a(ct, va, vb), va.shape).reshape(shape)))
    if tb:
        parents.append((b, lambda ct, va=va, vb=vb, shape=b.val.shape: _unbroadcast(back_b(ct, va, vb), vb.shape).reshape(shape)))
    return Tracer(out, parents, batched)


def asarray(x, dtype=None):
    dt = None if dtype is None else canonicalize_dtype(dtype)
    if isinstance(x, Tracer):
        if dt is None or dt == x.val.dtype:
            return x
        return Tracer(x.val.astype(dt), [(x, lambda ct, d=x.val.dtype: _np.asarray(ct).astype(d))], x.batched)
    v = _np.asarray(x, dtype=dt)
    if dt is None and v.dtype in (_np.float64, _np.int64):
        v = v.astype(canonicalize_dtype(v.dtype))
    return Tracer(v)


array = asarray


def ravel(x):
    if not isinstance(x, Tracer):
        return Tracer(_np.ravel(x))
    new = (x.val.shape[0], -1) if x.batched else (-1,)
    return Tracer(x.val.reshape(new), [(x, lambda ct, s=x.val.shape: _np.asarray(ct).reshape(s))], x.batched)


def concatenate(xs):
    xs = [x if isinstance(x, Tracer) else Tracer(_np.asarray(x)) for x in xs]
    batched = any(x.batched for x in xs)
    vals = []
    for x in xs:
        v = x.val
        if batched and not x.batched:
            n = next(y.val.shape[0] for y in xs if y.batched)
            v = _np.broadcast_to(_np.atleast_1d(v), (n,) + _np.atleast_1d(v).shape)
        elif batched and v.ndim == 1:
            v = v[:, None]
        elif not batched:
            v = _np.atleast_1d(v)
        vals.append(v)
    axis = 1 if batched else 0
    out = _np.concatenate(vals, axis=axis)
    parents, off = [], 0
    for x, v in zip(xs, vals):
        n = v.shape[axis]

        def back(ct, off=off, n=n, x=x, v=v):
            piece = _np.take(_np.asarray(ct), range(off, off + n), axis=axis)
            if batched and not x.batched:
                piece = piece.sum(axis=0)
            return piece.reshape(x.val.shape)

        parents.append((x, back))
        off += n
    return Tracer(out, parents, batched)


def exp(x):
    v = _np.exp(_unwrap(x))
    return Tracer(v, [(x, lambda ct, v=v: ct * v)] if isinstance(x, Tracer) else [], getattr(x, "batched", False))


def log(x):
    v = _unwrap(x)
    return Tracer(_np.log(v), [(x, lambda ct, v=v: ct / v)] if isinstance(x, Tracer) else [], getattr(x, "batched", False))


def sum(x):  # noqa: A001  (per example: every axis behind the batch axis)
    if not isinstance(x, Tracer):
        return Tracer(_np.sum(x))
    axes = tuple(range(1, x.val.ndim)) if x.batched else None

    def back(ct, shape=x.val.shape):
        ct = _np.asarray(ct)
        return _np.broadcast_to(ct.reshape(ct.shape + (1,) * (len(shape) - ct.ndim)), shape).copy()

    return Tracer(x.val.sum(axis=axes), [(x, back)], x.batched)


def zeros(shape, dtype=float64):
    return Tracer(_np.zeros(shape, dtype=canonicalize_dtype(dtype)))
