"""
`galois_amd.GF(...)`: the field class factory.

Mirrors galois.GF (reference: /root/reference/src/galois/_fields/_factory.py:53-302 and the prime / extension
factories :364-532): same call signatures, same defaults (smallest primitive root / Conway polynomial with 'x' as the
primitive element), same flyweight behaviour (identical arguments return the same class object), same dtype rules
(_domains/_meta.py:94-102, _fields/_ufunc.py:35-48, 97-111).  What differs is what a class carries: a gfa_field_t
handle (include/galois_amd.h) instead of Numba-JIT'd ufuncs.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib as L
from . import _numtheory as nt
from ._array import DTYPES, FieldArray, FieldArrayMeta

_CLASSES: dict[tuple, type] = {}


class IrreduciblePoly:
    """Minimal stand-in for the reference's Poly object of a field's irreducible polynomial: int(), str(), coeffs."""

    def __init__(self, value: int, p: int):
        self._value = int(value)
        self._p = int(p)
        self.coeffs = nt.poly_from_int(self._value, self._p)  # highest degree first
        self.degree = len(self.coeffs) - 1

    def __int__(self) -> int:
        return self._value

    def __index__(self) -> int:
        return self._value

    def __str__(self) -> str:
        return nt.poly_str(self.coeffs)

    def __repr__(self) -> str:
        return f"Poly({self}, GF({self._p}))"

    def __eq__(self, other) -> bool:
        return int(other) == self._value if isinstance(other, (int, IrreduciblePoly)) else NotImplemented

    def __hash__(self):
        return hash((self._value, self._p))


def _poly_like_to_int(poly, p: int, what: str) -> int:
    if isinstance(poly, (int, np.integer)):
        return int(poly)
    if isinstance(poly, IrreduciblePoly):
        return int(poly)
    if isinstance(poly, str):
        return _parse_poly_str(poly, p)
    if isinstance(poly, (list, tuple, np.ndarray)):
        return nt.poly_to_int([int(c) % p for c in poly], p)
    if hasattr(poly, "__int__"):
        return int(poly)
    raise TypeError(f"Argument {what!r} must be an int, str, or coefficient sequence, not {type(poly)}.")


def _parse_poly_str(s: str, p: int) -> int:
    value = 0
    for term in s.replace(" ", "").replace("-", "+-").split("+"):
        if not term:
            continue
        if "x" in term:
            c, _, e = term.partition("x")
            c = {"": 1, "-": -1}.get(c.rstrip("*"), None) if c.rstrip("*") in ("", "-") else int(c.rstrip("*"))
            e = int(e.lstrip("^").lstrip("**")) if e else 1
        else:
            c, e = int(term), 0
        value += (c % p) * p**e
    return value


def _determine_dtypes(order: int, restrict_squares: bool) -> list:
    """_domains/_meta.py:94-102 for GF(2^m); _fields/_ufunc.py:35-48, 97-111 for GF(p) and GF(p^m), p odd."""
    int64_max = np.iinfo(np.int64).max
    dtypes = [d for d in DTYPES if np.iinfo(d).max >= order - 1 and (not restrict_squares or int64_max >= (order - 1) ** 2)]
    return dtypes if dtypes else [np.object_]


def GF(*args, irreducible_poly=None, primitive_element=None, verify: bool = True, compile: str | None = None,
       repr: str | None = None):
    """
    Creates (or returns the cached) FieldArray subclass for GF(p^m).

        GF(order)  |  GF(characteristic, degree)

    Keyword arguments as galois.GF (_fields/_factory.py:53-302).  `compile` in {None, "auto", "jit-lookup",
    "jit-calculate"} selects the device kernel family.  `repr` other than None/"int" is not supported.
    """
    if len(args) == 1:
        order = args[0]
        if not isinstance(order, (int, np.integer)):
            raise TypeError(f"Argument 'order' must be an instance of int, not {type(order)}.")
        order = int(order)
        if order < 2:
            raise ValueError(f"Argument 'order' must be a prime power, not {order}.")
        p, m = nt.prime_power(order)
    elif len(args) == 2:
        p, m = args
        if not isinstance(p, (int, np.integer)) or not isinstance(m, (int, np.integer)):
            raise TypeError("Arguments 'characteristic' and 'degree' must be integers.")
        p, m = int(p), int(m)
        if not nt.is_prime(p):
            raise ValueError(f"Argument 'characteristic' must be prime, not {p}.")
        if m < 1:
            raise ValueError(f"Argument 'degree' must be at least 1, not {m}.")
    else:
        raise TypeError("Argument '*args' must be of the form 'order' or 'characteristic, degree'.")
    big = p**m > 2**128 or (p**m == 2**128 and p != 2)
    if big:
        # Orders above 2^128: k limbs per element (galois_amd/_big.py, up to 1024 bits).  The defaults of smaller fields need the
        # factorisation of q - 1 (primitive root / element searches, primitivity of the polynomial), which is out of reach for a
        # 200-bit integer in general (the reference's own defaults would not return either): the caller names both.
        from ._big import limbs_for

        limbs_for(p**m)  # NotImplementedError beyond 1024 bits
        if primitive_element is None or (m > 1 and irreducible_poly is None):
            raise ValueError(
                f"GF({p}^{m}) has order > 2^128: pass `primitive_element=` (and `irreducible_poly=` for an extension field) explicitly; "
                "the default searches factor q - 1, which is not feasible at this size."
            )
        if verify:
            raise ValueError(f"GF({p}^{m}) has order > 2^128: pass `verify=False` (verifying a primitive element factors q - 1).")
    if compile is not None and compile not in ("auto", "jit-lookup", "jit-calculate"):
        raise ValueError(
            f"Argument 'compile' must be in ['auto', 'jit-lookup', 'jit-calculate'], not {compile!r} "
            "(there is no Python/CPU arithmetic mode in galois_amd)."
        )
    if repr not in (None, "int"):
        raise NotImplementedError("Only the integer element representation is implemented.")

    if m == 1:
        if irreducible_poly is not None:
            raise ValueError(
                "Argument 'irreducible_poly' can only be specified for extension fields, not the prime field "
                f"GF({p})."
            )
        alpha = nt.primitive_root(p) if primitive_element is None else int(primitive_element)
        if not 0 < alpha < p:
            raise ValueError(f"Argument 'primitive_element' must be non-zero in the field 0 < x < {p}, not {alpha}.")
        key = (p, 1, alpha, 0)
        if key not in _CLASSES:
            if verify and primitive_element is not None and not nt.is_primitive_root(alpha, p):
                raise ValueError(f"Argument 'primitive_element' must be a primitive root modulo {p}, {alpha} is not.")
            irr_int = 2 * p - alpha  # f(x) = x - alpha (_factory.py:405)
            _CLASSES[key] = _make_class(p, 1, irr_int, alpha, True, None)
    else:
        prime_subfield = GF(p) if p < 2**64 or not big else None  # (a prime subfield above 2^128 has no default generator either)
        is_primitive_poly = None
        verify_poly = verify
        verify_element = verify
        if irreducible_poly is None:
            irr_int = nt.conway_poly(p, m)
            is_primitive_poly = True
            verify_poly = False  # Conway polynomials are irreducible and primitive (_factory.py:447-456)
            if primitive_element is None:
                alpha = p  # the polynomial 'x'
                verify_element = False
        else:
            irr_int = _poly_like_to_int(irreducible_poly, p, "irreducible_poly")
        coeffs = nt.poly_from_int(irr_int, p)
        if len(coeffs) - 1 != m:
            raise ValueError(f"Argument 'irreducible_poly' must have degree equal to {m}, not {len(coeffs) - 1}.")
        if primitive_element is not None:
            alpha = _poly_like_to_int(primitive_element, p, "primitive_element")
            if not 0 < alpha < p**m:
                raise ValueError(f"Argument 'primitive_element' must have degree strictly less than {m}.")
        elif irreducible_poly is not None:
            alpha = _default_primitive_element(irr_int, p, verify_poly)
            verify_element = False
        key = (p, m, alpha, irr_int)
        if key not in _CLASSES:
            if verify_poly and not nt.is_irreducible(coeffs, p):
                raise ValueError(f"Argument 'irreducible_poly' must be irreducible, {nt.poly_str(coeffs)} is not.")
            if verify_element and not nt.is_primitive_element(nt.poly_from_int(alpha, p), coeffs, p):
                raise ValueError(
                    f"Argument 'primitive_element' must be a multiplicative generator of GF({p}^{m}), "
                    f"{nt.poly_str(nt.poly_from_int(alpha, p))} is not."
                )
            if is_primitive_poly is None:
                is_primitive_poly = False if big else nt.is_primitive_element([1, 0], coeffs, p)  # (unknown above 2^128: not computed)
            _CLASSES[key] = _make_class(p, m, irr_int, alpha, is_primitive_poly, prime_subfield)
    field = _CLASSES[key]
    if compile is not None:
        field.compile(compile)
    return field


_DEFAULT_ALPHA: dict[tuple[int, int], int] = {}


def _default_primitive_element(irr_int: int, p: int, verify_poly: bool) -> int:
    """galois.primitive_element(irreducible_poly): the smallest primitive element (_factory.py:462-465)."""
    key = (irr_int, p)
    if key not in _DEFAULT_ALPHA:
        coeffs = nt.poly_from_int(irr_int, p)
        if verify_poly and not nt.is_irreducible(coeffs, p):
            raise ValueError(f"Argument 'irreducible_poly' must be irreducible, {nt.poly_str(coeffs)} is not.")
        _DEFAULT_ALPHA[key] = nt.primitive_element(coeffs, p)
    return _DEFAULT_ALPHA[key]


def _make_class(p: int, m: int, irr_int: int, alpha: int, is_primitive_poly: bool, prime_subfield) -> type:
    order = p**m
    if order > 2**128 or (order == 2**128 and p != 2):
        return _make_big_class(p, m, irr_int, alpha, is_primitive_poly, prime_subfield)
    if order >= 2**64:
        return _make_wide_class(p, m, irr_int, alpha, is_primitive_poly, prime_subfield)
    coeffs = nt.poly_from_int(irr_int, p)
    handle = ctypes.c_void_p()
    arr = (ctypes.c_uint64 * (m + 1))(*coeffs) if m > 1 else None
    L.check(L.lib().gfa_field_create(p, m, arr, alpha, ctypes.byref(handle)), f"GF({p}^{m})")
    dtypes = _determine_dtypes(order, restrict_squares=not (p == 2 and m > 1))
    object_dtype = dtypes == [np.object_]
    lookup_ok = order <= 2**20
    name = f"FieldArray_{p}_{alpha}" if m == 1 else f"FieldArray_{p}_{m}_{alpha}_{irr_int}"
    ns = {
        "_characteristic": p,
        "_degree": m,
        "_order": order,
        "_irreducible_poly": IrreduciblePoly(irr_int, p),
        "_primitive_element_int": alpha,
        "_primitive_element_str": nt.poly_str(nt.poly_from_int(alpha, p)) if m > 1 else str(alpha),
        "_is_primitive_poly": bool(is_primitive_poly),
        "_prime_subfield": prime_subfield,
        "_dtypes": dtypes,
        "_object_dtype": object_dtype,
        "_ufunc_modes": (["jit-lookup", "jit-calculate"] if lookup_ok else ["jit-calculate"]),
        "_handle": handle,
        "__module__": __name__,
        "__hash__": None,
    }
    cls = FieldArrayMeta(name, (FieldArray,), ns)
    mode = L.lib().gfa_field_get_mode(handle)
    cls._default_ufunc_mode = "jit-lookup" if mode == L.MODE_LOOKUP else "jit-calculate"
    return cls


def _make_wide_class(p: int, m: int, irr_int: int, alpha: int, is_primitive_poly: bool, prime_subfield) -> type:
    """2^64 <= order <= 2^128: two 64-bit limbs per element on the device (galois_amd/_wide.py, csrc/gfa_wide.hip)."""
    from ._wide import WideFieldArray, wide_params

    kind, words = wide_params(p, m, irr_int)
    handle = ctypes.c_void_p()
    arr = (ctypes.c_uint64 * 27)(*words)
    L.check(L.lib().gfa_wfield_create(kind, m, arr, ctypes.byref(handle)), f"GF({p}^{m})")
    name = f"FieldArray_{p}_{alpha}" if m == 1 else f"FieldArray_{p}_{m}_{alpha}_{irr_int}"
    ns = {
        "_characteristic": p,
        "_degree": m,
        "_order": p**m,
        "_irreducible_poly": IrreduciblePoly(irr_int, p),
        "_primitive_element_int": alpha,
        "_primitive_element_str": nt.poly_str(nt.poly_from_int(alpha, p)) if m > 1 else str(alpha),
        "_is_primitive_poly": bool(is_primitive_poly),
        "_prime_subfield": prime_subfield,
        "_dtypes": [np.object_],
        "_object_dtype": True,
        # the reference runs these fields in "python-calculate"; here the same formulas run in device kernels
        "_ufunc_modes": ["jit-calculate"],
        "_default_ufunc_mode": "jit-calculate",
        "_handle": None,
        "_wide_handle": handle,
        "__module__": __name__,
        "__hash__": None,
    }
    cls = FieldArrayMeta(name, (WideFieldArray,), ns)
    import weakref

    weakref.finalize(cls, L.lib().gfa_wfield_destroy, handle)  # the class owns its device-side descriptor
    return cls


def _make_big_class(p: int, m: int, irr_int: int, alpha: int, is_primitive_poly: bool, prime_subfield) -> type:
    """order > 2^128: 4, 8 or 16 limbs of 64 bits per element on the device (galois_amd/_big.py, csrc/gfa_big.hip)."""
    from ._big import BigFieldArray, big_params, limbs_for

    nl = limbs_for(p**m)
    kind, words = big_params(p, m, irr_int, nl)
    handle = ctypes.c_void_p()
    arr = (ctypes.c_uint64 * 113)(*words)
    L.check(L.lib().gfa_bfield_create(kind, m, nl, arr, ctypes.byref(handle)), f"GF({p}^{m})")
    name = f"FieldArray_{p}_{alpha}" if m == 1 else f"FieldArray_{p}_{m}_{alpha}_{irr_int}"
    ns = {
        "_characteristic": p,
        "_degree": m,
        "_order": p**m,
        "_irreducible_poly": IrreduciblePoly(irr_int, p),
        "_primitive_element_int": alpha,
        "_primitive_element_str": nt.poly_str(nt.poly_from_int(alpha, p)) if m > 1 else str(alpha),
        "_is_primitive_poly": bool(is_primitive_poly),
        "_prime_subfield": prime_subfield,
        "_dtypes": [np.object_],
        "_object_dtype": True,
        "_ufunc_modes": ["jit-calculate"],
        "_default_ufunc_mode": "jit-calculate",
        "_handle": None,
        "_wide_handle": handle,
        "_NL": nl,
        "__module__": __name__,
        "__hash__": None,
    }
    cls = FieldArrayMeta(name, (BigFieldArray,), ns)
    import weakref

    weakref.finalize(cls, L.lib().gfa_bfield_destroy, handle)  # the class owns its device-side descriptor
    return cls


def Field(*args, **kwargs):
    """Deprecated alias of GF, kept because the reference still uses it internally (_fields/_factory.py:330-361)."""
    import warnings

    warnings.warn("galois_amd.Field() is deprecated; use galois_amd.GF() instead.", DeprecationWarning, stacklevel=2)
    return GF(*args, **kwargs)
