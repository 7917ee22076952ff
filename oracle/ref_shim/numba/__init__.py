"""
TEST INFRASTRUCTURE ONLY -- a ~40-line stand-in for the `numba` package.

The reference (mhostetter/galois, mounted read-only at /root/reference) is pure Python but imports
numba at module scope.  numba/llvmlite are not installed in this container, so this module lets the
reference's *own* scalar Python functions execute unmodified under CPython ("python-calculate"
mode, /root/reference/src/galois/_domains/_ufunc.py:146-159).  Nothing here re-implements numba:
`jit` returns the function unchanged, `vectorize` wraps it in np.frompyfunc.

Used only by oracle/ref_shim/load_reference.py to (a) validate oracle/gf_oracle.c and (b) generate
the golden fixtures under tests/golden/.  Never imported by the product package.
"""
import types as _types

import numpy as np

__version__ = "0.59.0"
prange = range


def jit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        fn = args[0]
        fn.py_func = fn
        return fn

    def deco(fn):
        fn.py_func = fn
        return fn

    return deco


njit = jit


def vectorize(sigs=None, **kwargs):
    def deco(fn):
        nin = sigs[0].count("int64") - 1 if sigs else fn.__code__.co_argcount
        # numba freezes module globals (CHARACTERISTIC, EXP, LOG, MULTIPLY, ...) at compile time
        # (/root/reference/src/galois/_domains/_ufunc.py:109-116 sets them right before compiling).
        # Emulate that by binding the scalar function to a snapshot of its globals, otherwise the
        # next field to "compile" would overwrite the constants of this one.
        frozen = _types.FunctionType(fn.__code__, dict(fn.__globals__), fn.__name__, fn.__defaults__, fn.__closure__)
        return np.frompyfunc(frozen, nin, 1)

    return deco


class _Type:
    def __init__(self, name):
        self.name = name

    def __getitem__(self, item):
        return _Type(f"{self.name}[]")

    def __call__(self, *args):
        return _Type(f"{self.name}({', '.join(getattr(a, 'name', str(a)) for a in args)})")


int64 = _Type("int64")
uint64 = _Type("uint64")
bool_ = _Type("bool")

from . import types  # noqa: E402,F401
