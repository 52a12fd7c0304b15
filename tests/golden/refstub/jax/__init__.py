"""NumPy/SciPy-backed stand-in for the handful of `jax` entry points the reference's
hot-path modules touch.  TEST INFRASTRUCTURE ONLY: it exists so that the *unmodified*
reference files under /root/reference can be imported in the build container (where jax is
not installable) to generate golden vectors.  It is never imported by the product package.

Everything runs in float64 (NumPy default), i.e. the equivalent of JAX_ENABLE_X64=1.
"""
import functools as _ft

from . import numpy  # noqa: F401
from . import lax  # noqa: F401
from . import random  # noqa: F401
from . import tree_util  # noqa: F401
from . import scipy  # noqa: F401


def jit(fun=None, **_kw):
    """Identity decorator; accepts `static_argnames=` both directly and via functools.partial."""
    if fun is None:
        return lambda f: f
    return fun


def vmap(fun, in_axes=0, out_axes=0):
    import numpy as _np

    @_ft.wraps(fun)
    def mapped(*args):
        n = len(args[0])
        return numpy.array([fun(*[a[i] for a in args]) for i in range(n)])

    return mapped


def value_and_grad(*_a, **_k):  # pragma: no cover - autodiff is not emulated
    raise NotImplementedError("the stub has no autodiff; gradients are checked by finite differences")
