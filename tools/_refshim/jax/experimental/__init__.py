from . import sparse  # noqa: F401
