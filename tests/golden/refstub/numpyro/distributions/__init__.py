from . import constraints  # noqa: F401
from . import util  # noqa: F401


class Distribution:
    arg_constraints = {}

    def __init__(self, batch_shape=(), event_shape=(), validate_args=None):
        self.batch_shape = batch_shape
        self.event_shape = event_shape


class _Named(Distribution):
    def __init__(self, *a, **k):
        self.args, self.kwargs = a, k
        super().__init__()


class Gamma(_Named):
    pass


class Categorical(_Named):
    pass


class Normal(_Named):
    pass


class Uniform(_Named):
    pass


class HalfNormal(_Named):
    pass
