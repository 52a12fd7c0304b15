import numpy as _np
import scipy.special as _sp

from ..numpy import _wrap


def betaln(a, b):
    return _wrap(_np.asarray(_sp.betaln(a, b)))


def erf(x):
    return _wrap(_np.asarray(_sp.erf(x)))


def logsumexp(a, axis=None, b=None, keepdims=False):
    with _np.errstate(all="ignore"):
        return _wrap(_np.asarray(_sp.logsumexp(a, axis=axis, b=b, keepdims=keepdims)))
