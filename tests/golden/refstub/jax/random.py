"""Minimal PRNG facade (only enough for imports and the seeded draws the golden script makes)."""
import numpy as _np


def PRNGKey(seed):
    return _np.array([0, int(seed)], dtype=_np.uint32)


def _rng(key):
    return _np.random.default_rng(int(key[0]) * 2**32 + int(key[1]))


def split(key, num=2):
    r = _rng(key)
    return [_np.array(r.integers(0, 2**32, size=2), dtype=_np.uint32) for _ in range(num)]


def uniform(key, shape=(), minval=0.0, maxval=1.0, **_k):
    return _rng(key).uniform(minval, maxval, size=shape)


def normal(key, shape=(), **_k):
    return _rng(key).normal(size=shape)


def choice(key, a, shape=(), replace=True, p=None, **_k):
    return _rng(key).choice(a, size=shape if shape != () else None, replace=replace, p=p)
