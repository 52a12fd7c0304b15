"""A MINIMAL stand-in for the parts of JAX that gwinferno_amd's custom_vjp adapter touches (likelihood._evaluate_jax):
``custom_vjp`` / ``defvjp``, ``pure_callback``, ``ShapeDtypeStruct`` and a handful of ``jax.numpy`` functions over an
opaque ``Tracer`` value.  TEST INFRASTRUCTURE for tests/test_jax_adapter_cpu.py only (JAX is not installable in the build
or test images): it checks the adapter's plumbing -- shapes declared to pure_callback, primal / forward agreement, what
the backward rule returns -- not JAX itself.  No autodiff here: the test pulls cotangents through the recorded rule."""
import numpy as _np

from . import numpy  # noqa: F401  (jax.numpy)
from .numpy import Tracer, _unwrap, _wrap


class ShapeDtypeStruct:
    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), _np.dtype(dtype)


def _check(result, spec, where):
    arr = _np.asarray(result)
    if arr.shape != spec.shape:
        raise TypeError(f"pure_callback {where}: host function returned shape {arr.shape}, declared {spec.shape}")
    if arr.dtype != spec.dtype:
        raise TypeError(f"pure_callback {where}: host function returned dtype {arr.dtype}, declared {spec.dtype}")
    return _wrap(arr)


CALLBACK_CALLS = []


def pure_callback(host, result_shape_dtypes, *args, vmap_method=None):
    """Calls ``host`` with concrete NumPy arrays and verifies the result against the declared structure (``vmap_method`` is
    what a batching rule would consult; nothing is batched here)."""
    out = host(*[_unwrap(a) for a in args])
    CALLBACK_CALLS.append(host)
    if isinstance(result_shape_dtypes, (tuple, list)):
        if not isinstance(out, (tuple, list)) or len(out) != len(result_shape_dtypes):
            raise TypeError("pure_callback: result structure differs from the declared one")
        return tuple(_check(o, s, f"output {i}") for i, (o, s) in enumerate(zip(out, result_shape_dtypes)))
    return _check(out, result_shape_dtypes, "output")


class custom_vjp:
    """Records the forward / backward rules.  Calling the function runs BOTH the primal body and the forward rule (they
    must agree, as JAX requires) and keeps the residuals so that a test can pull cotangents back with :meth:`pull`."""

    last = None

    def __init__(self, fun):
        self.fun, self.fwd, self.bwd = fun, None, None

    def defvjp(self, fwd, bwd):
        self.fwd, self.bwd = fwd, bwd

    def __call__(self, *args):
        if self.fwd is None:
            raise RuntimeError("custom_vjp called before defvjp")
        primal = self.fun(*args)
        out, residuals = self.fwd(*args)
        flat_p = primal if isinstance(primal, tuple) else (primal,)
        flat_o = out if isinstance(out, tuple) else (out,)
        if len(flat_p) != len(flat_o) or any(not _np.array_equal(_unwrap(a), _unwrap(b), equal_nan=True) for a, b in zip(flat_p, flat_o)):
            raise AssertionError("custom_vjp: the forward rule's outputs differ from the primal function's")
        custom_vjp.last = (self, residuals, out, args)
        return out

    @classmethod
    def pull(cls, cotangents):
        """Apply the recorded backward rule of the most recent call to ``cotangents`` (same structure as the outputs)."""
        self, residuals, out, args = cls.last
        grads = self.bwd(residuals, cotangents)
        if not isinstance(grads, tuple) or len(grads) != len(args):
            raise TypeError("custom_vjp: the backward rule must return one cotangent per primal argument")
        return tuple(_unwrap(g) for g in grads)


def jit(fun=None, **_):
    return fun if fun is not None else (lambda f: f)


__all__ = ["ShapeDtypeStruct", "pure_callback", "custom_vjp", "jit", "Tracer"]
