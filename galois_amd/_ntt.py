"""
galois_amd.ntt / intt and the np.fft.fft / np.fft.ifft override on field arrays.

Host-side mirror of the reference front ends (paths relative to /root/reference/src/galois):
  * ntt, intt, _ntt .................. _ntt.py:16-278  (argument checks, default modulus, norm bookkeeping)
  * fft_jit.__call__ / ifft_jit ....... _domains/_function.py:177-212 (pad/truncate to n, omega = alpha^((q-1)/n) or its
                                        inverse, optional division by n)
The transform itself is gfa_ntt (include/galois_amd.h) on the device.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from ._array import FieldArray, _ptr, _stream
from ._numtheory import is_prime


def _field_fft(x: FieldArray, n=None, axis=-1, norm=None, inverse: bool = False) -> FieldArray:
    """np.fft.fft / np.fft.ifft on a 1-D field array (fft_jit.__call__, _domains/_function.py:177-212)."""
    if not isinstance(x, FieldArray):
        raise TypeError(f"Argument 'x' must be a field array, not {type(x)}.")
    norm = "backward" if norm is None else norm
    if not axis == -1:
        raise ValueError("The FFT is only implemented on 1-D arrays.")
    if norm not in ["forward", "backward"]:
        raise ValueError("DFT normalization can only be applied to the forward or backward transform, not 'ortho'.")
    if x.ndim != 1:
        raise ValueError("The FFT is only implemented on 1-D arrays.")
    scale = norm == ("backward" if inverse else "forward")
    if type(x)._limbed:  # order >= 2^64: mixed-radix recursion on the two-limb kernels (galois_amd/_wide.py)
        return x._fft(n, inverse, scale)
    return _transform_rows(x, n, inverse, scale)


def _transform_rows(x: FieldArray, n, inverse: bool, scale: bool) -> FieldArray:
    """Transforms every row (last axis) of `x`; 1-D input = one row.  Rows are padded / truncated to n."""
    cls = type(x)
    t = x._t
    length = t.shape[-1]
    if n is None:
        n = length
    n = int(n)
    if n < length:
        t = t[..., :n]
    elif n > length:
        pad = torch.zeros(tuple(t.shape[:-1]) + (n - length,), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad], dim=-1)
    t = t.contiguous()
    omega = cls.primitive_root_of_unity(n)  # ValueError if n does not divide q - 1 (_fields/_array.py:1182-1185)
    if inverse:
        omega = cls._scalar(L.OP_RECIP, omega)
    out = torch.empty_like(t)
    batch = t.numel() // n if n else 0
    L.check(L.lib().gfa_ntt(cls._handle, _ptr(t), _ptr(out), n, batch, omega, 1 if scale else 0, x._gfa_dtype(), _stream()),
            "gfa_ntt")
    return cls._wrap(out, x._np_dtype)


def fft_batched(x: FieldArray, inverse: bool = False, scaled: bool | None = None) -> FieldArray:
    """Device extension (not in the reference, whose FFT is 1-D only): independent transforms of every row of a 2-D
    array in one launch.  `scaled` defaults to the reference's convention (inverse divides by n)."""
    if x.ndim != 2:
        raise ValueError("fft_batched expects a 2-D array (batch, n).")
    scale = inverse if scaled is None else bool(scaled)
    return _transform_rows(x, None, inverse, scale)


def _field_convolve(a: FieldArray, b: FieldArray, mode: str = "full") -> FieldArray:
    """np.convolve on 1-D field arrays = polynomial multiplication (convolve_jit.__call__, _domains/_function.py:116-130).

    Short products run in one direct kernel (gfa_convolve).  Long products over a prime field whose multiplicative
    group has enough 2-adicity run as three NTTs and a pointwise product -- the direct consumer of the fast transform
    (SURVEY.md section 8(f) item 1).  Long products over any other prime field below 2^32 are taken inside gfa_convolve
    through three auxiliary NTT primes and the Chinese remainder theorem (gfa_conv_crt.hip); the result is the same exact
    polynomial product every way."""
    if not isinstance(a, FieldArray) or not isinstance(b, FieldArray) or type(a) is not type(b):
        raise TypeError(f"Arguments of 'convolve' must be arrays over the same field, not {type(a)} and {type(b)}.")
    if not mode == "full":
        raise ValueError(f"Operation 'convolve' currently only supports mode of 'full', not {mode!r}.")
    if a.ndim != 1 or b.ndim != 1 or a.size == 0 or b.size == 0:
        raise ValueError("Operation 'convolve' requires non-empty 1-D arrays.")
    cls = type(a)
    if cls._limbed:  # order >= 2^64: two limbs per element, direct kernel (gfa_wide_convolve)
        return a._convolve(b)
    na, nb = a.size, b.size
    n_out = na + nb - 1
    n_fft = 1 << (n_out - 1).bit_length()
    if cls.is_prime_field and min(na, nb) >= 64 and n_out >= 2048 and (cls.order - 1) % n_fft == 0 and n_fft <= 2**28:
        both = cls.Zeros((2, n_fft), dtype=a.dtype if a.dtype != np.dtype(object) else None)
        both._t[0, :na] = a._t
        both._t[1, :nb] = a._same_storage(b)
        spec = _transform_rows(both, None, inverse=False, scale=False)
        prod = spec[0] * spec[1]
        full = _transform_rows(prod.reshape(1, n_fft), None, inverse=True, scale=True)
        return cls._wrap(full._t[0, :n_out].contiguous(), a._np_dtype)
    ta = a._t.contiguous()
    tb = a._same_storage(b).contiguous()
    out = torch.empty(n_out, dtype=ta.dtype, device=ta.device)
    L.check(L.lib().gfa_convolve(cls._handle, _ptr(ta), na, _ptr(tb), nb, _ptr(out), a._gfa_dtype(), _stream()), "gfa_convolve")
    return cls._wrap(out, a._np_dtype)


def _max_value(x) -> int:
    if isinstance(x, FieldArray):
        return int(x.numpy().max()) if x.size else 0
    arr = np.asarray(x, dtype=object if not isinstance(x, np.ndarray) else None)
    return int(max(int(v) for v in np.asarray(arr).ravel()))


def _default_modulus(max_x: int, size: int) -> int:
    """The prime modulus galois.ntt picks when none is given: the first prime m * size + 1 with m >= ceil(max(x) / size)
    (_ntt.py:250-254, same floating-point ceiling)."""
    m = int(np.ceil(max_x / size))
    while not is_prime(m * size + 1):
        m += 1
    return m * size + 1


def _ntt(x, size=None, modulus=None, forward=True, scaled=True):
    from ._factory import GF

    for name, val in (("size", size), ("modulus", modulus)):
        if val is not None and not isinstance(val, (int, np.integer)):
            raise TypeError(f"Argument {name!r} must be an instance of int, not {type(val)}.")
    if not isinstance(scaled, (bool, np.bool_)):
        raise TypeError(f"Argument 'scaled' must be an instance of bool, not {type(scaled)}.")
    if size is None:
        size = len(x)
    size = int(size)
    max_x = _max_value(x)
    if modulus is None:
        modulus = _default_modulus(max_x, size)
    modulus = int(modulus)
    if not size >= len(x):
        raise ValueError(f"Argument 'size' must be at least the length of the input which is {len(x)}, not {size}.")
    if not is_prime(modulus):
        raise ValueError(f"Argument 'modulus' must be prime, {modulus} is not.")
    if not (modulus - 1) % size == 0:
        raise ValueError("Argument 'modulus' must equal m * size + 1, where 'size' is the size of the NTT transform.")
    if not modulus > max_x:
        raise ValueError(f"Argument 'modulus' must be at least the max value of the input which is {max_x}, not {modulus}.")
    field = GF(modulus)
    if isinstance(x, FieldArray) and type(x) is not field:
        x = x.numpy()
    xf = x if isinstance(x, FieldArray) else field(np.asarray(x, dtype=object) if modulus > 2**63 else x)
    if forward:
        return _field_fft(xf, n=size, inverse=False)
    return _field_fft(xf, n=size, inverse=True, norm="backward" if scaled else "forward")


def ntt(x, size: int | None = None, modulus: int | None = None) -> FieldArray:
    """Number-theoretic transform of x over GF(p) (galois.ntt, _ntt.py:16-118)."""
    if not isinstance(x, (tuple, list, np.ndarray, FieldArray)):
        raise TypeError(f"Argument 'x' must be an instance of (tuple, list, np.ndarray, FieldArray), not {type(x)}.")
    if isinstance(x, FieldArray) and not type(x).is_prime_field:
        raise ValueError(f"If argument 'x' is a FieldArray, it must be a prime field, not {type(x)}.")
    if modulus is None and isinstance(x, FieldArray):
        modulus = type(x).characteristic
    return _ntt(x, size=size, modulus=modulus, forward=True)


def intt(X, size: int | None = None, modulus: int | None = None, scaled: bool = True) -> FieldArray:
    """Inverse number-theoretic transform (galois.intt, _ntt.py:122-236)."""
    if not isinstance(X, (tuple, list, np.ndarray, FieldArray)):
        raise TypeError(f"Argument 'X' must be an instance of (tuple, list, np.ndarray, FieldArray), not {type(X)}.")
    if isinstance(X, FieldArray) and not type(X).is_prime_field:
        raise ValueError(f"If argument 'X' is a FieldArray, it must be a prime field, not {type(X)}.")
    if modulus is None and isinstance(X, FieldArray):
        modulus = type(X).characteristic
    return _ntt(X, size=size, modulus=modulus, forward=False, scaled=scaled)
