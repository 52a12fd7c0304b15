from . import autoguide  # noqa: F401


class SVI:
    pass


class Trace_ELBO:
    pass


class NUTS:
    pass


class HMC:
    pass


class MCMC:
    pass
