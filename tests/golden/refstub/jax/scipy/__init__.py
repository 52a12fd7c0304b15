from . import integrate  # noqa: F401
from . import special  # noqa: F401
