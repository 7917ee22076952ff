"""Stand-in for numba.types (TEST INFRASTRUCTURE ONLY; see numba/__init__.py in this directory)."""
from . import _Type, int64, bool_  # noqa: F401


class FunctionType:
    def __init__(self, signature):
        self.signature = signature


def Array(dtype, ndim, layout, readonly=False):
    return _Type(f"Array({dtype.name},{ndim},{layout})")


def Tuple(items):
    return _Type("Tuple")
