"""`numpyro` stand-in (TEST INFRASTRUCTURE ONLY, see ../jax/__init__.py): primitives record the
deterministic / factor sites of one model execution into `SITES`; `sample` returns a fixed value
injected through `SAMPLE_VALUES` (only `unscaled_rate` is drawn inside the reference hot path)."""
from . import distributions  # noqa: F401
from . import infer  # noqa: F401
from . import optim  # noqa: F401

SITES = {}
SAMPLE_VALUES = {}


def reset():
    SITES.clear()


def deterministic(name, value):
    SITES[name] = value
    return value


def factor(name, value):
    SITES[name] = value
    return value


def sample(name, fn=None, *a, **k):
    if name in SAMPLE_VALUES:
        return SAMPLE_VALUES[name]
    raise KeyError(f"stub numpyro.sample: no value provided for site {name!r}")


class plate:
    def __init__(self, name, size, **k):
        self.size = size

    def __enter__(self):
        import numpy as np

        return np.arange(self.size)

    def __exit__(self, *a):
        return False


def set_host_device_count(n):
    return None
