"""
galois_amd.ReedSolomon / galois_amd.BCH -- host-side mirror of the reference's cyclic-code front ends over the device
kernels.

Reference (paths relative to /root/reference/src/galois):
  * ReedSolomon.__init__ .............. _codes/_reed_solomon.py:111-218 (argument checks, default field GF(2^m) with
                                        matlab_primitive_poly, alpha = primitive n-th root of unity, roots, g(x), H)
  * _LinearCode.encode/detect/decode .. _codes/_linear.py:58-186 (1-D vs 2-D, shortened codes, erasures mask, outputs)
  * _check_and_convert_* .............. _codes/_linear.py:202-251
  * _convert_codeword_to_message ...... _codes/_cyclic.py:129-149
  * BCH.__init__ ...................... _codes/_bch.py:106-240 (+ _generator_poly_from_d / _from_k :1178-1252)
  * _CyclicCode.__init__ .............. _codes/_cyclic.py:26-54 (parity-check polynomial, G, H)
The arithmetic (generator polynomial roots, parity matrix, encode, syndromes, Berlekamp-Massey/Chien/Forney) is behind
gfa_rs_* / gfa_bch_create in include/galois_amd.h.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib as L
from . import _numtheory as nt
from ._array import FieldArray, _GFA_DTYPE, _TORCH_STORAGE, _ptr, _stream, _to_storage
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


class _CyclicCode:
    """encode / detect / decode front end shared by ReedSolomon and BCH (_LinearCode, _codes/_linear.py:58-251)."""

    _handle = None

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                L.lib().gfa_rs_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def _describe(self):
        """Host copies of what the library derived: roots, generator polynomial, systematic parity matrix."""
        n, k, nk = self._n, self._k, self._n - self._k
        nroots = self._d - 1
        roots = np.zeros(max(nroots, 1), dtype=np.uint64)
        g = np.zeros(nk + 1, dtype=np.uint64)
        P = np.zeros(max(k * nk, 1), dtype=np.uint64)
        L.check(L.lib().gfa_rs_describe(self._handle, roots.ctypes.data_as(L._u64p), g.ctypes.data_as(L._u64p),
                                        P.ctypes.data_as(L._u64p)), "gfa_rs_describe")
        self._roots = roots[:nroots].astype(np.int64)
        self._generator_poly = _HostPoly(g, self._field)
        self._P = P[: k * nk].reshape(k, nk).astype(np.int64)
        self._G = None
        self._H = None

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

    _ext_order = property(lambda self: self._field.order)

    # ---- helpers ----------------------------------------------------------------------------------------------
    def _to_u8_device(self, x, what: str):
        """array-like / FieldArray -> (device tensor in the code's symbol storage, original FieldArray for dtype bookkeeping).

        Codes whose syndrome field has at most 256 elements run on the byte kernels (uint8 symbols).  Larger codes --
        RS(1023, k) over GF(2^10), BCH(1023, k) over GF(2), ... -- keep the narrowest storage that holds a symbol
        (uint8 / uint16 / uint32) and run on the table-driven kernels of gfa_rs_wide.hip (syndrome fields up to 2^20)."""
        arr = x if isinstance(x, FieldArray) and type(x) is self.field else self.field(x)
        if self._ext_order > 2**20:
            raise NotImplementedError(f"{what}: the device path covers codes whose syndrome field has order <= 2^20.")
        width = 1 if (self._ext_order <= 256 or self.field.order <= 256) else (2 if self.field.order <= 65536 else 4)
        want = _TORCH_STORAGE[width]
        if arr._t.dtype != want:
            return _to_storage(arr._t, want), arr
        return arr._t, arr

    def _verify_decoded(self, out: torch.Tensor):
        """Hook for BCH: the reference views the decoder's integer output as the symbol field (_bch.py:1300)."""

    def _wrap(self, t_sym: torch.Tensor, like: FieldArray) -> FieldArray:
        if like._t.dtype != t_sym.dtype:
            return self.field._wrap(_to_storage(t_sym, like._t.dtype), like._np_dtype)
        return self.field._wrap(t_sym, like._np_dtype)

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
        out = torch.empty((N, nk if parity_only else ks + nk), dtype=m2.dtype, device=m2.device)
        if out.numel():  # the identity code (n == k) has no parity symbols
            L.check(L.lib().gfa_rs_encode(self._handle, _ptr(m2), ks, _ptr(out), N, 1 if parity_only else 0,
                                          _GFA_DTYPE[m2.element_size()], _stream()), "gfa_rs_encode")
        if is_1d:
            out = out[0]
        return self._wrap(out, like)

    # ---- detect (_linear.py:95-117) ---------------------------------------------------------------------------
    def detect(self, codeword):
        t, _ = self._to_u8_device(codeword, "detect")
        t2, is_1d = self._check_codeword(t)
        N, ns = t2.shape
        det = torch.empty(N, dtype=torch.uint8, device=t2.device)
        L.check(L.lib().gfa_rs_detect(self._handle, _ptr(t2), ns, _ptr(det), N, _GFA_DTYPE[t2.element_size()], _stream()),
                "gfa_rs_detect")
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
            if isinstance(erasures, torch.Tensor):  # a mask that already lives on the device stays there
                if erasures.dtype != torch.bool:
                    raise TypeError(f"Argument 'erasures' must have dtype bool, not {erasures.dtype}.")
                if tuple(erasures.shape) != tuple(t.shape):
                    raise ValueError(f"Argument 'erasures' must have shape {tuple(t.shape)}, not {tuple(erasures.shape)}.")
                er_t = erasures.to(t2.device).reshape(N, ns).to(torch.uint8).contiguous()
            else:
                er = np.asarray(erasures)
                if er.dtype != bool:
                    raise TypeError(f"Argument 'erasures' must have dtype bool, not {er.dtype}.")
                if er.shape != tuple(t.shape):
                    raise ValueError(f"Argument 'erasures' must have shape {tuple(t.shape)}, not {er.shape}.")
                er_t = torch.from_numpy(np.ascontiguousarray(er.reshape(N, ns)).astype(np.uint8)).to(t2.device)
        out = torch.empty_like(t2)
        nerr = torch.empty(N, dtype=torch.int64, device=t2.device)
        L.check(L.lib().gfa_rs_decode(self._handle, _ptr(t2), _ptr(er_t) if er_t is not None else None, ns, _ptr(out),
                                      _ptr(nerr), N, _GFA_DTYPE[t2.element_size()], _stream()), "gfa_rs_decode")
        self._verify_decoded(out)
        if output == "message":
            ks = self.k - (self.n - ns)
            dec = torch.empty((N, ks), dtype=out.dtype, device=out.device)  # _cyclic.py:129-138
            L.check(L.lib().gfa_rs_extract_message(self._handle, _ptr(out), ns, _ptr(dec), N, _GFA_DTYPE[out.element_size()],
                                                   _stream()), "gfa_rs_extract_message")
        else:
            dec = out
        n_errors = nerr.cpu().numpy()
        if is_1d:
            dec, n_errors = dec[0], int(n_errors[0])
        result = self._wrap(dec, like)
        if errors:
            return result, n_errors
        return result


class ReedSolomon(_CyclicCode):
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
        self._describe()

    def __repr__(self) -> str:
        return f"<Reed-Solomon Code: [{self.n}, {self.k}, {self.d}] over {self.field.name}>"

    def __str__(self) -> str:
        return (
            f"Reed-Solomon Code:\n  [n, k, d]: [{self.n}, {self.k}, {self.d}]\n  field: {self.field.name}\n"
            f"  generator_poly: {self.generator_poly}\n  is_primitive: {self.is_primitive}\n"
            f"  is_narrow_sense: {self.is_narrow_sense}\n  is_systematic: {self.is_systematic}"
        )

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


class BCH(_CyclicCode):
    """A general BCH(n, k) code over GF(p), p prime, with syndrome arithmetic in GF(p^m); see galois.BCH."""

    def __init__(self, n: int, k: int | None = None, d: int | None = None, field=None, extension_field=None, alpha=None,
                 c: int = 1, systematic: bool = True):
        for name, val, opt in (("n", n, False), ("k", k, True), ("d", d, True), ("c", c, False)):
            if not (opt and val is None) and (not isinstance(val, (int, np.integer)) or isinstance(val, bool)):
                raise TypeError(f"Argument {name!r} must be an instance of int, not {type(val)}.")
        for name, val in (("field", field), ("extension_field", extension_field)):
            if val is not None and not (isinstance(val, type) and issubclass(val, FieldArray)):
                raise TypeError(f"Argument {name!r} must be a subclass of FieldArray, not {val!r}.")
        if not isinstance(systematic, bool):
            raise TypeError(f"Argument 'systematic' must be an instance of bool, not {type(systematic)}.")
        if d is not None and not d >= 1:
            raise ValueError(f"Argument 'd' must be at least 1, not {d}.")
        if not c >= 0:
            raise ValueError(f"Argument 'c' must be at least 0, not {c}.")
        n = int(n)
        if field is None:
            field = GF(2)
        if not field.is_prime_field:
            raise ValueError(
                "Current BCH codes over GF(q) for prime power q are not supported. "
                "Proper Galois field towers are needed first."
            )
        q = field.order
        if extension_field is None:
            m = _ilog(n, q) + 1
            assert q ** (m - 1) < n + 1 <= q**m
            extension_field = GF(q**m, irreducible_poly=nt.matlab_primitive_poly(q, m)) if m > 1 else field
        if extension_field.characteristic != q:
            raise ValueError(f"Argument 'extension_field' must be an extension of {field.name}, not {extension_field.name}.")
        if alpha is None:
            alpha_int = extension_field.primitive_root_of_unity(n)
        else:
            alpha_int = int(alpha)
            if not 0 < alpha_int < extension_field.order:
                raise ValueError(f"Argument 'alpha' must be a non-zero element of {extension_field.name}.")
        self._field, self._extension_field = field, extension_field
        self._n, self._alpha_int, self._c = n, alpha_int, int(c)
        if d is not None:
            g, roots = self._generator_poly_from_d(int(d))
            kk = n - (len(g) - 1)
            if k not in [None, kk]:
                raise ValueError(
                    f"The requested [{n}, {k}, {d}] code is not consistent. "
                    f"When designing the code with design distance {d}, the resulting code is [{n}, {kk}, {d}]."
                )
            k = kk
        elif k is not None:
            g, roots = self._generator_poly_from_k(int(k))
            d = len(roots) + 1
        else:
            raise ValueError("Argument 'k' or 'd' must be provided to define the code size.")
        k, d = int(k), int(d)
        if not 1 <= k <= n:
            raise ValueError(f"Argument 'k' must be in [1, n], not {k}.")
        self._k, self._d = k, d
        self._is_systematic = systematic
        self._is_primitive = n == extension_field.order - 1
        self._is_narrow_sense = c == 1
        h, rem = _pmod_divmod([1] + [0] * (n - 1) + [q - 1], g, q)  # h(x) = (x^n - 1) / g(x)  (_cyclic.py:45-49)
        assert rem == [0]
        self._parity_check_poly = _HostPoly(h, field)
        garr = np.array(g, dtype=np.uint64)
        handle = ctypes.c_void_p()
        L.check(L.lib().gfa_bch_create(extension_field._handle, q, n, k, d, self._c, alpha_int, garr.ctypes.data_as(L._u64p),
                                       1 if systematic else 0, ctypes.byref(handle)), "BCH")
        self._handle = handle
        self._describe()
        assert [int(v) for v in self._roots] == roots

    # ---- generator polynomial (host integers; element arithmetic through gfa_scalar) ---------------------------------
    def _minimal_poly(self, beta: int) -> list:
        """Minimal polynomial over GF(p) of beta in GF(p^m): product of (x - conjugate) over the distinct Frobenius
        conjugates (FieldArray.minimal_poly, _fields/_array.py:1979-2050).  Coefficients highest degree first."""
        E, p = self._extension_field, self._field.order
        conj, x = [], int(beta)
        while x not in conj:
            conj.append(x)
            x = E._scalar(L.OP_POW, x, p)
        poly = [1]
        for r in conj:
            nr = E._scalar(L.OP_NEG, r)
            nxt = poly + [0]
            for i, cf in enumerate(poly):
                nxt[i + 1] = E._scalar(L.OP_ADD, nxt[i + 1], E._scalar(L.OP_MUL, cf, nr))
            poly = nxt
        assert all(cf < p for cf in poly)
        return poly

    def _generator_poly_from_d(self, d: int):
        """_bch.py:1178-1197: g(x) = product of the distinct minimal polynomials of alpha^c .. alpha^(c+d-2)."""
        E, p = self._extension_field, self._field.order
        roots = [E._scalar(L.OP_POW, self._alpha_int, self._c + i) for i in range(d - 1)]
        g, seen = [1], []
        for r in roots:
            mi = self._minimal_poly(r)
            if mi not in seen:
                g = _pmod_mul(g, mi, p)
                seen.append(mi)
        return g, roots

    def _generator_poly_from_k(self, k: int):
        """_bch.py:1200-1252: binary search for a design distance of that size, then the largest such d."""
        n = self._n
        m = _ilog(self._extension_field.order, self._field.order)
        possible_d = list(range((n - k) // m + 1, (n - k) + 2))
        while len(possible_d) > 0:
            idx = len(possible_d) // 2
            d = possible_d[idx]
            g, roots = self._generator_poly_from_d(d)
            if len(g) - 1 < n - k:
                possible_d = possible_d[idx + 1:]
            elif len(g) - 1 == n - k:
                break
            else:
                possible_d = possible_d[0:idx]
        else:
            raise ValueError(
                f"The BCH({n}, {k}) code over {self._field.name} with alpha={self._alpha_int} and c={self._c} does not exist."
            )
        best = (g, roots)
        while True:
            d += 1
            g, roots = self._generator_poly_from_d(d)
            if len(g) - 1 == n - k:
                best = (g, roots)
            elif len(g) - 1 > n - k:
                break
        return best

    def __repr__(self) -> str:
        return f"<BCH Code: [{self.n}, {self.k}, {self.d}] over {self.field.name}>"

    def __str__(self) -> str:
        return (
            f"BCH Code:\n  [n, k, d]: [{self.n}, {self.k}, {self.d}]\n  field: {self.field.name}\n"
            f"  extension_field: {self.extension_field.name}\n  generator_poly: {self.generator_poly}\n"
            f"  is_primitive: {self.is_primitive}\n  is_narrow_sense: {self.is_narrow_sense}\n"
            f"  is_systematic: {self.is_systematic}"
        )

    extension_field = property(lambda self: self._extension_field)
    parity_check_poly = property(lambda self: self._parity_check_poly)
    _ext_order = property(lambda self: self._extension_field.order)

    @property
    def H(self) -> np.ndarray:
        """Parity-check matrix: the non-systematic generator matrix of the reversed h(x) (_cyclic.py:222-226)."""
        if self._H is None:
            hrev = [int(v) for v in self._parity_check_poly.coeffs[::-1]]
            Hm = np.zeros((self.n - self.k, self.n), dtype=np.int64)
            for i in range(self.n - self.k):
                Hm[i, i : i + len(hrev)] = hrev
            self._H = Hm
        return self._H

    def _verify_decoded(self, out: torch.Tensor):
        # dec_codeword.view(self.field) -> _verify_array_values (_bch.py:1300, _fields/_array.py:170-177): a miscorrection
        # whose Forney values fall outside GF(p) leaves symbols >= p in the decoder's integer output
        wide = out if out.dtype == torch.uint8 else (out.to(torch.int64) & ((1 << (8 * out.element_size())) - 1))
        if self._field.order < (1 << (8 * out.element_size())) and bool((wide >= self._field.order).any()):
            bad = wide[wide >= self._field.order]
            raise ValueError(
                f"{self._field.name} arrays must have elements in `0 <= x < {self._field.order}`, "
                f"not {bad.cpu().numpy()}."
            )


def _pmod_trim(a):
    a = list(a)
    while len(a) > 1 and a[0] == 0:
        a.pop(0)
    return a


def _pmod_mul(a, b, p):
    """Product in GF(p)[x], coefficients highest degree first."""
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                out[i + j] = (out[i + j] + x * y) % p
    return _pmod_trim(out)


def _pmod_divmod(a, b, p):
    a, b = _pmod_trim(a), _pmod_trim(b)
    if len(a) < len(b):
        return [0], a
    inv = pow(b[0], -1, p)
    a = list(a)
    quo = []
    for i in range(len(a) - len(b) + 1):
        cf = a[i] * inv % p
        quo.append(cf)
        if cf:
            for j, y in enumerate(b):
                a[i + j] = (a[i + j] - cf * y) % p
    return _pmod_trim(quo), _pmod_trim(a[len(a) - len(b) + 1:] or [0])


def _ilog(n: int, b: int) -> int:
    """Largest e with b**e <= n (galois.ilog)."""
    e = 0
    while b ** (e + 1) <= n:
        e += 1
    return e
