from . import struct  # noqa: F401
