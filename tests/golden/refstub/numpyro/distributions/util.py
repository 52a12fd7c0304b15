import numpy as _np


def validate_sample(fn):
    return fn


def promote_shapes(*args, shape=()):
    return [(_np.asarray(a) if not _np.isscalar(a) else a) for a in args]


def is_prng_key(key):
    return True
