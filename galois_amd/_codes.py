"""
galois_amd.ReedSolomon -- host-side mirror of the reference's Reed-Solomon front end over the device kernels.

Reference (paths relative to /root/reference/src/galois):
  * ReedSolomon.__init__ .............. _codes/_reed_solomon.py:111-218 (argument checks, default field GF(2^m) with
                                        matlab_primitive_poly, alpha = primitive n-th root of unity, roots, g(x), H)
  * _LinearCode.encode/detect/decode .. _codes/_linear.py:58-186 (1-D vs 2-D, shortened codes, erasures mask, outputs)
  * _check_and_convert_* .............. _codes/_linear.py:202-251
  * _convert_codeword_to_message ...... _codes/_cyclic.py:129-149
The arithmetic (generator polynomial, parity matrix, encode, syndromes, Berlekamp-Massey/Chien/Forney) is behind
gfa_rs_* in include/galois_amd.h.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib as L
from . import _numtheory as nt
from ._array import FieldArray, _ptr, _stream
from ._factory import GF


class _HostPoly:
    """Coefficients (highest degree first) of a polynomial over the code's field, as host integers."""

    def __init__(self, coeffs, field):
        self.coeffs = np.array([int(c) for c in coeffs], dtype=np.int64)
        self.field = field
        self.degree = len(self.coeffs) - 1

    def __str__(self) -> str:
        return nt.poly_str([int(c) for c in self.coeffs])

    def __repr__(self) -> str:
        return f"Poly({self}, {self.field.name})"

    def __int__(self) -> int:
        return nt.poly_to_int([int(c) for c in self.coeffs], self.field.order)


class ReedSolomon:
    """A general RS(n, k) code over GF(q); see galois.ReedSolomon for the full contract."""

    def __init__(self, n: int, k: int | None = None, d: int | None = None, field=None, alpha=None, c: int = 1,
                 systematic: bool = True):
        for name, val, opt in (("n", n, False), ("k", k, True), ("d", d, True), ("c", c, False)):
            if not (opt and val is None) and (not isinstance(val, (int, np.integer)) or isinstance(val, bool)):
                raise TypeError(f"Argument {name!r} must be an instance of int, not {type(val)}.")
        if field is not None and not (isinstance(field, type) and issubclass(field, FieldArray)):
            raise TypeError(f"Argument 'field' must be a subclass of FieldArray, not {field!r}.")
        if not isinstance(systematic, bool):
            raise TypeError(f"Argument 'systematic' must be an instance of bool, not {type(systematic)}.")
        if d is not None and not d >= 1:
            raise ValueError(f"Argument 'd' must be at least 1, not {d}.")
        if not c >= 0:
            raise ValueError(f"Argument 'c' must be at least 0, not {c}.")
        n = int(n)
        if field is None:
            q = 2
            m = _ilog(n, q) + 1
            assert q ** (m - 1) < n + 1 <= q**m
            field = GF(q**m, irreducible_poly=nt.matlab_primitive_poly(q, m)) if m > 1 else GF(2)
        if alpha is None:
            alpha_int = field.primitive_root_of_unity(n)
        else:
            alpha_int = int(alpha)
            if not 0 < alpha_int < field.order:
                raise ValueError(f"Argument 'alpha' must be a non-zero element of {field.name}.")
        if d is not None and k is not None:
            if not d == n - k + 1:
                raise ValueError(
                    "Arguments 'k' and 'd' were provided but are inconsistent. For Reed-Solomon codes, d = n - k + 1."
                )
        elif d is not None:
            k = n - (d - 1)
        elif k is not None:
            d = (n - k) + 1
        else:
            raise ValueError("Argument 'k' or 'd' must be provided to define the code size.")
        k, d = int(k), int(d)
        if not 1 <= k <= n:
            raise ValueError(f"Argument 'k' must be in [1, n], not {k}.")

        self._field = field
        self._n, self._k, self._d = n, k, d
        self._alpha_int = alpha_int
        self._c = int(c)
        self._is_systematic = systematic
        self._is_primitive = n == field.order - 1
        self._is_narrow_sense = c == 1
        handle = ctypes.c_void_p()
        L.check(L.lib().gfa_rs_create(field._handle, n, k, self._c, alpha_int, 1 if systematic else 0,
                                      ctypes.byref(handle)), "ReedSolomon")
        self._handle = handle
        nk = n - k
        roots = np.zeros(max(nk, 1), dtype=np.uint64)
        g = np.zeros(nk + 1, dtype=np.uint64)
        P = np.zeros(max(k * nk, 1), dtype=np.uint64)
        L.check(L.lib().gfa_rs_describe(handle, roots.ctypes.data_as(L._u64p), g.ctypes.data_as(L._u64p),
                                        P.ctypes.data_as(L._u64p)), "gfa_rs_describe")
        self._roots = roots[:nk].astype(np.int64)
        self._generator_poly = _HostPoly(g, field)
        self._P = P[: k * nk].reshape(k, nk).astype(np.int64)
        self._G = None
        self._H = None

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                L.lib().gfa_rs_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def __repr__(self) -> str:
        return f"<Reed-Solomon Code: [{self.n}, {self.k}, {self.d}] over {self.field.name}>"

    def __str__(self) -> str:
        return (
            f"Reed-Solomon Code:\n  [n, k, d]: [{self.n}, {self.k}, {self.d}]\n  field: {self.field.name}\n"
            f"  generator_poly: {self.generator_poly}\n  is_primitive: {self.is_primitive}\n"
            f"  is_narrow_sense: {self.is_narrow_sense}\n  is_systematic: {self.is_systematic}"
        )

    # ---- properties -------------------------------------------------------------------------------------------
    field = property(lambda self: self._field)
    n = property(lambda self: self._n)
    k = property(lambda self: self._k)
    d = property(lambda self: self._d)
    t = property(lambda self: (self._d - 1) // 2)
    c = property(lambda self: self._c)
    alpha = property(lambda self: self._alpha_int)
    roots = property(lambda self: self._roots)
    generator_poly = property(lambda self: self._generator_poly)
    is_primitive = property(lambda self: self._is_primitive)
    is_narrow_sense = property(lambda self: self._is_narrow_sense)
    is_systematic = property(lambda self: self._is_systematic)

    @property
    def G(self) -> np.ndarray:
        """Generator matrix (k x n) as host integers (_cyclic.py:198-226)."""
        if self._G is None:
            if self._is_systematic:
                self._G = np.hstack([np.eye(self.k, dtype=np.int64), self._P])
            else:
                G = np.zeros((self.k, self.n), dtype=np.int64)
                for i in range(self.k):
                    G[i, i : i + self._generator_poly.degree + 1] = self._generator_poly.coeffs
                self._G = G
        return self._G

    @property
    def H(self) -> np.ndarray:
        """Parity-check matrix: np.power.outer(roots, arange(n-1, -1, -1)) (_reed_solomon.py:218), host integers."""
        if self._H is None:
            H = np.zeros((self.n - self.k, self.n), dtype=np.int64)
            for i, r in enumerate(self._roots):
                for j in range(self.n):
                    H[i, j] = self.field._scalar(L.OP_POW, int(r), self.n - 1 - j)
            self._H = H
        return self._H

    # ---- helpers ----------------------------------------------------------------------------------------------
    def _to_u8_device(self, x, what: str):
        """array-like / FieldArray -> (uint8 device tensor, original FieldArray for dtype bookkeeping)."""
        arr = x if isinstance(x, FieldArray) and type(x) is self.field else self.field(x)
        if arr._t.element_size() != 1:
            if self.field.order > 256:
                raise NotImplementedError(f"{what}: the device path covers codes over fields of order <= 256.")
            return arr._t.to(torch.uint8), arr
        return arr._t, arr

    def _wrap(self, t_u8: torch.Tensor, like: FieldArray) -> FieldArray:
        if like._t.element_size() != 1:
            return self.field._wrap(t_u8.to(like._t.dtype), like._np_dtype)
        return self.field._wrap(t_u8, like._np_dtype)

    # ---- encode (_linear.py:58-93) -----------------------------------------------------------------------------
    def encode(self, message, output: str = "codeword") -> FieldArray:
        if output not in ["codeword", "parity"]:
            raise ValueError(f"Argument 'output' must be in ['codeword', 'parity'], not {output!r}.")
        if output == "parity" and not self.is_systematic:
            raise ValueError("Argument 'output' may only be 'parity' for systematic codes.")
        t, like = self._to_u8_device(message, "encode")
        if t.dim() > 2:
            raise ValueError(f"Argument 'message' can be either 1-D or 2-D, not {t.dim()}-D.")
        if t.dim() == 0 or not 1 <= t.shape[-1] <= self.k:
            raise ValueError(
                f"Argument 'message' must be a 1-D or 2-D array with last dimension between 1 and {self.k}, "
                f"not shape {tuple(t.shape)}."
            )
        is_1d = t.dim() == 1
        m2 = t.reshape(1, -1) if is_1d else t
        m2 = m2.contiguous()
        N, ks = m2.shape
        nk = self.n - self.k
        parity_only = output == "parity"
        out = torch.empty((N, nk if parity_only else ks + nk), dtype=torch.uint8, device=m2.device)
        L.check(L.lib().gfa_rs_encode(self._handle, _ptr(m2), ks, _ptr(out), N, 1 if parity_only else 0, L.U8, _stream()),
                "gfa_rs_encode")
        if is_1d:
            out = out[0]
        return self._wrap(out, like)

    # ---- detect (_linear.py:95-117) ---------------------------------------------------------------------------
    def detect(self, codeword):
        t, _ = self._to_u8_device(codeword, "detect")
        t2, is_1d = self._check_codeword(t)
        N, ns = t2.shape
        det = torch.empty(N, dtype=torch.uint8, device=t2.device)
        L.check(L.lib().gfa_rs_detect(self._handle, _ptr(t2), ns, _ptr(det), N, L.U8, _stream()), "gfa_rs_detect")
        detected = det.cpu().numpy().astype(bool)
        return bool(detected[0]) if is_1d else detected

    def _check_codeword(self, t: torch.Tensor):
        if t.dim() == 0 or t.dim() > 2 or not self.n - self.k + 1 <= t.shape[-1] <= self.n:
            raise ValueError(
                f"Argument 'codeword' must be a 1-D or 2-D array with last dimension between {self.n - self.k + 1} "
                f"and {self.n}, not shape {tuple(t.shape)}."
            )
        is_1d = t.dim() == 1
        t2 = (t.reshape(1, -1) if is_1d else t).contiguous()
        return t2, is_1d

    # ---- decode (_linear.py:137-186) --------------------------------------------------------------------------
    def decode(self, codeword, erasures=None, output: str = "message", errors: bool = False):
        if output not in ["message", "codeword"]:
            raise ValueError(f"Argument 'output' must be in ['message', 'codeword'], not {output!r}.")
        t, like = self._to_u8_device(codeword, "decode")
        t2, is_1d = self._check_codeword(t)
        N, ns = t2.shape
        er_t = None
        if erasures is not None:
            er = erasures.cpu().numpy() if isinstance(erasures, torch.Tensor) else np.asarray(erasures)
            if er.dtype != bool:
                raise TypeError(f"Argument 'erasures' must have dtype bool, not {er.dtype}.")
            if er.shape != tuple(t.shape):
                raise ValueError(f"Argument 'erasures' must have shape {tuple(t.shape)}, not {er.shape}.")
            er_t = torch.from_numpy(np.ascontiguousarray(er.reshape(N, ns)).astype(np.uint8)).to(t2.device)
        out = torch.empty_like(t2)
        nerr = torch.empty(N, dtype=torch.int64, device=t2.device)
        L.check(L.lib().gfa_rs_decode(self._handle, _ptr(t2), _ptr(er_t) if er_t is not None else None, ns, _ptr(out),
                                      _ptr(nerr), N, L.U8, _stream()), "gfa_rs_decode")
        if output == "message":
            ks = self.k - (self.n - ns)
            dec = torch.empty((N, ks), dtype=torch.uint8, device=out.device)  # _cyclic.py:129-138
            L.check(L.lib().gfa_rs_extract_message(self._handle, _ptr(out), ns, _ptr(dec), N, L.U8, _stream()),
                    "gfa_rs_extract_message")
        else:
            dec = out
        n_errors = nerr.cpu().numpy()
        if is_1d:
            dec, n_errors = dec[0], int(n_errors[0])
        result = self._wrap(dec, like)
        if errors:
            return result, n_errors
        return result


def _ilog(n: int, b: int) -> int:
    """Largest e with b**e <= n (galois.ilog)."""
    e = 0
    while b ** (e + 1) <= n:
        e += 1
    return e
