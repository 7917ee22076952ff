"""
galois_amd.Poly -- a small dense univariate polynomial over a device field: the part of the reference's Poly that is
array-sized work.  Paths relative to /root/reference/src/galois:

  * Poly.__call__ (element-wise and square-matrix evaluation) ... _polys/_poly.py:862-950 over evaluate_elementwise_jit /
                                                                 evaluate_matrix_jit (_polys/_dense.py:404-470)
  * + - * (polynomial and scalar), negation ..................... _polys/_dense.py:54-123 (multiply = np.convolve)
Everything symbolic (factoring, gcd, irreducibility tests, sparse/binary representations) stays out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from . import _numtheory as nt
from ._array import FieldArray, _ptr, _stream


class Poly:
    """Dense polynomial with coefficients in degree-descending order, held as a 1-D device field array."""

    def __init__(self, coeffs, field=None, order: str = "desc"):
        if order not in ["desc", "asc"]:
            raise ValueError(f"Argument 'order' must be in ['desc', 'asc'], not {order!r}.")
        if isinstance(coeffs, FieldArray):
            if field is not None and type(coeffs) is not field:
                raise TypeError(f"Argument 'coeffs' is over {type(coeffs).name} but 'field' is {field.name}.")
            c = coeffs
        else:
            if field is None:
                from ._factory import GF

                field = GF(2)
            arr = np.asarray(coeffs)
            if arr.dtype != object and np.issubdtype(arr.dtype, np.integer) and arr.size and arr.min() < 0:
                arr = arr % field.characteristic if field.is_prime_field else arr  # "-1" style coefficients over GF(p)
            c = field(arr)
        if c.ndim != 1 or c.size == 0:
            raise ValueError(f"Argument 'coeffs' must be a non-empty 1-D array, not shape {tuple(c.shape)}.")
        if order == "asc":
            c = np.flip(c)
        self._field = type(c)
        self._coeffs = self._trim(c)

    @staticmethod
    def _trim(c: FieldArray) -> FieldArray:
        nz = torch.nonzero(c._t.reshape(c.size, -1).any(dim=1))  # (fields of order >= 2^64: two limbs per element)
        if nz.numel() == 0:
            return c[-1:] if c.size else c
        return c[int(nz[0].item()):]

    field = property(lambda self: self._field)
    coeffs = property(lambda self: self._coeffs)
    degree = property(lambda self: self._coeffs.size - 1)

    def __repr__(self) -> str:
        return f"Poly({self}, {self._field.name})"

    def __str__(self) -> str:
        return nt.poly_str([int(v) for v in self._coeffs.numpy()])

    def __eq__(self, other) -> bool:
        if not isinstance(other, Poly) or other._field is not self._field or other.degree != self.degree:
            return False
        return bool(np.all(self._coeffs == other._coeffs))

    # ---- evaluation --------------------------------------------------------------------------------------------------
    def __call__(self, x, elementwise: bool = True):
        F = self._field
        xa = x if isinstance(x, FieldArray) and type(x) is F else F(x)
        if F._limbed:  # order >= 2^64: two-limb kernels (gfa_wide_poly_evaluate; matrix Horner on gfa_wide_matmul)
            if elementwise:
                return self._coeffs._poly_evaluate(xa)
            if not (xa.ndim == 2 and xa.shape[0] == xa.shape[1]):
                raise ValueError(f"Argument 'x' must be a square matrix when evaluating the polynomial not element-wise, not shape {tuple(xa.shape)}.")
            eye = F.Identity(xa.shape[0])
            acc = eye * self._coeffs[0]
            for j in range(1, self._coeffs.size):
                acc = acc @ xa + eye * self._coeffs[j]
            return acc
        c = xa._same_storage(self._coeffs).contiguous()
        if elementwise:
            t = xa._t.contiguous()
            out = torch.empty_like(t)
            L.check(L.lib().gfa_poly_evaluate(F._handle, _ptr(c), c.numel(), _ptr(t), _ptr(out), t.numel(), xa._gfa_dtype(),
                                              _stream()), "gfa_poly_evaluate")
            return F._wrap(out, xa._np_dtype)
        # matrix evaluation: Horner with matrix products (evaluate_matrix_jit, _dense.py:443-470)
        if not (xa.ndim == 2 and xa.shape[0] == xa.shape[1]):
            raise ValueError(f"Argument 'x' must be a square matrix when evaluating the polynomial not element-wise, not shape {tuple(xa.shape)}.")
        eye = F.Identity(xa.shape[0], dtype=xa.dtype if xa.dtype != np.dtype(object) else None)
        cs = F._wrap(c, xa._np_dtype)
        acc = eye * cs[0]
        for j in range(1, cs.size):
            acc = acc @ xa + eye * cs[j]
        return acc

    # ---- arithmetic --------------------------------------------------------------------------------------------------
    def _coerce(self, other) -> "Poly":
        if isinstance(other, Poly):
            if other._field is not self._field:
                raise TypeError(f"Both polynomials must be over the same field, not {self._field.name} and {other._field.name}.")
            return other
        if isinstance(other, FieldArray) and type(other) is self._field and other.ndim == 0:
            return Poly(other.reshape(1))
        raise TypeError(f"Cannot combine a polynomial over {self._field.name} with {type(other)}.")

    def _aligned(self, other: "Poly"):
        a, b = self._coeffs, other._coeffs
        n = max(a.size, b.size)
        F = self._field
        if F._limbed:
            pad = lambda v: np.concatenate([F.Zeros(n - v.size), v]) if v.size < n else v
            return pad(a), pad(b)
        ta = torch.zeros(n, dtype=a._t.dtype, device=a._t.device)
        tb = torch.zeros(n, dtype=a._t.dtype, device=a._t.device)
        ta[n - a.size:] = a._t
        tb[n - b.size:] = a._same_storage(b)
        return F._wrap(ta, a._np_dtype), F._wrap(tb, a._np_dtype)

    def __add__(self, other):
        a, b = self._aligned(self._coerce(other))
        return Poly(a + b)

    __radd__ = __add__

    def __sub__(self, other):
        a, b = self._aligned(self._coerce(other))
        return Poly(a - b)

    def __rsub__(self, other):
        return self._coerce(other) - self

    def __neg__(self):
        return Poly(-self._coeffs)

    def __mul__(self, other):
        if isinstance(other, (int, np.integer)):
            return Poly(self._coeffs * int(other))  # scalar (repeated-addition) multiplication, _poly.py:1406-1416
        o = self._coerce(other)
        return Poly(np.convolve(self._coeffs, o._coeffs))

    __rmul__ = __mul__

    def derivative(self, k: int = 1) -> "Poly":
        """Formal derivative (_polys/_poly.py:1098-1160): coefficient of x^(j-1) is (j mod p) * a_j."""
        if not isinstance(k, (int, np.integer)) or k < 1:
            raise ValueError(f"Argument 'k' must be a positive integer, not {k}.")
        p = self
        for _ in range(int(k)):
            if p.degree == 0:
                return Poly(p._field([0]))
            c = p._coeffs[:-1]
            mult = np.arange(p.degree, 0, -1)
            p = Poly(c * mult)
        return p


def berlekamp_massey(sequence, output: str = "minimal") -> Poly:
    """galois.berlekamp_massey (_lfsr.py:1560-1625): the minimal (characteristic) polynomial c(x) of a linear recurrent
    sequence, or its connection polynomial C(x) = reverse of c(x).  The LFSR-object outputs of the reference ("fibonacci",
    "galois") are out of scope."""
    if not isinstance(sequence, FieldArray):
        raise TypeError(f"Argument 'sequence' must be a FieldArray, not {type(sequence)}.")
    if not isinstance(output, str):
        raise TypeError(f"Argument 'output' must be a string, not {type(output)}.")
    if not sequence.ndim == 1:
        raise ValueError(f"Argument 'sequence' must be 1-D, not {sequence.ndim}-D.")
    if output not in ["minimal", "connection", "fibonacci", "galois"]:
        raise ValueError(f"Argument 'output' must be in ['minimal', 'connection', 'fibonacci', 'galois'], not {output!r}.")
    if output in ["fibonacci", "galois"]:
        raise NotImplementedError("LFSR objects are outside this engine; use output='minimal' or 'connection'.")
    F = type(sequence)
    t = sequence._t.contiguous()
    n = t.numel()
    out = torch.empty_like(t)
    ln = torch.empty(1, dtype=torch.int64, device=t.device)
    L.check(L.lib().gfa_berlekamp_massey(F._handle, _ptr(t), n, 1, _ptr(out), _ptr(ln), sequence._gfa_dtype(), _stream()),
            "gfa_berlekamp_massey")
    asc = out[: int(ln.item())]
    if output == "connection":
        return Poly(F._wrap(torch.flip(asc, dims=(0,)).contiguous(), sequence._np_dtype))  # degree-descending C(x)
    return Poly(F._wrap(asc.contiguous(), sequence._np_dtype))  # c(x) = x^L C(1/x): the ascending C read as descending
