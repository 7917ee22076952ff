"""
TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/gf_oracle.c (the CPU restatement of the reference).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
package galois_amd/ never does (tests/test_no_oracle_in_product.py enforces it).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libgf_oracle.so")

ADD, SUB, MUL, DIV, NEG, RECIP, POW = range(7)
OK, ZERO_DIVISION, BAD_ARG = 0, 1, 2

_u64p = ctypes.POINTER(ctypes.c_uint64)
_i64p = ctypes.POINTER(ctypes.c_int64)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_u32p = ctypes.POINTER(ctypes.c_uint32)


def build(force: bool = False) -> str:
    """Compiles oracle/gf_oracle.c with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "gf_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.gfo_field_new.restype = ctypes.c_void_p
        L.gfo_field_new.argtypes = [ctypes.c_uint64, ctypes.c_uint64, _u64p, ctypes.c_uint64, ctypes.c_int]
        L.gfo_field_free.argtypes = [ctypes.c_void_p]
        L.gfo_field_set_lookup.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.gfo_field_tables.argtypes = [ctypes.c_void_p, _i64p, _i64p, _i64p, _i64p]
        L.gfo_ufunc.argtypes = [ctypes.c_void_p, ctypes.c_int, _u64p, ctypes.c_int64, _u64p, ctypes.c_int64, _u64p,
                                ctypes.c_int64]
        L.gfo_ufunc_u8.argtypes = [ctypes.c_void_p, ctypes.c_int, _u8p, _u8p, _u8p, ctypes.c_int64]
        L.gfo_ntt.argtypes = [ctypes.c_void_p, _u64p, ctypes.c_int64, ctypes.c_uint64, _i64p, ctypes.c_int64, _u64p]
        L.gfo_ntt_u32_pow2.argtypes = [ctypes.c_uint64, _u32p, ctypes.c_int64, ctypes.c_uint64, _u32p]
        L.gfo_matmul.argtypes = [ctypes.c_void_p, _u64p, _u64p, _u64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
        L.gfo_matmul.restype = None
        L.gfo_poly_eval.argtypes = [ctypes.c_void_p, _u64p, ctypes.c_int64, _u64p, ctypes.c_int64, _u64p]
        L.gfo_poly_eval.restype = None
        L.gfo_convolve.argtypes = [ctypes.c_void_p, _u64p, ctypes.c_int64, _u64p, ctypes.c_int64, _u64p]
        L.gfo_convolve.restype = None
        L.gfo_berlekamp_massey.argtypes = [ctypes.c_void_p, _u64p, ctypes.c_int64, _u64p, _i64p]
        L.gfo_berlekamp_massey.restype = None
        L.gfo_rs_decode.argtypes = [ctypes.c_void_p, _u64p, _u8p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                    ctypes.c_uint64, ctypes.c_int64, _u64p, ctypes.c_int64, _u64p, _i64p]
        L.gfo_rs_construct.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64, ctypes.c_int64,
                                       _u64p, _u64p, _u64p]
        L.gfo_rs_construct.restype = None
        L.gfo_rs_encode_u8.argtypes = [ctypes.c_void_p, _u8p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_int64, _u8p, _u8p]
        L.gfo_rs_decode_u8.argtypes = [ctypes.c_void_p, _u8p, _u8p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_uint64, ctypes.c_int64, _u64p, ctypes.c_int64, _u8p, _i64p]
        _lib = L
    return _lib


def _p(arr, typ):
    return arr.ctypes.data_as(typ)


def _as_u64(x):
    """Accepts ints / object arrays / integer arrays (values < 2^64) and returns a C-contiguous uint64 array."""
    a = np.asarray(x)
    if a.dtype == object:
        flat = [int(v) for v in a.ravel()]
        out = np.array(flat, dtype=np.uint64).reshape(a.shape)
    else:
        out = a.astype(np.uint64)
    return np.ascontiguousarray(out)


def poly_int_to_digits(poly_int: int, p: int, m: int) -> list[int]:
    """Integer representation (base p) of a degree-m polynomial -> m+1 coefficients, highest degree first."""
    digits = []
    for _ in range(m + 1):
        digits.append(poly_int % p)
        poly_int //= p
    return digits[::-1]


class OracleField:
    """One finite field GF(p^m) in the oracle, defined exactly as the reference defines a field class:
    (characteristic, degree, irreducible polynomial, primitive element) -- _fields/_factory.py:398-410, 501-514."""

    def __init__(self, p: int, m: int = 1, irreducible_poly: int | None = None, primitive_element: int | None = None,
                 lookup: bool = False):
        self.p, self.m, self.q = int(p), int(m), int(p) ** int(m)
        if m > 1:
            assert irreducible_poly is not None
            digits = poly_int_to_digits(int(irreducible_poly), p, m)
        else:
            digits = [1, 0]
        assert primitive_element is not None
        self.irreducible_poly = irreducible_poly
        self.alpha = int(primitive_element)
        arr = (ctypes.c_uint64 * len(digits))(*digits)
        self._h = lib().gfo_field_new(self.p, self.m, arr, self.alpha, 1 if lookup else 0)
        if not self._h:
            raise ValueError("gfo_field_new failed (bad field definition or primitive element is not a generator)")
        self.lookup = bool(lookup)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().gfo_field_free(self._h)
                self._h = None
        except Exception:
            pass

    def set_lookup(self, flag: bool):
        lib().gfo_field_set_lookup(self._h, 1 if flag else 0)

    def tables(self):
        q = self.q
        E = np.zeros(2 * q, dtype=np.int64)
        L = np.zeros(q, dtype=np.int64)
        Z = np.zeros(q, dtype=np.int64)
        ze = ctypes.c_int64(0)
        lib().gfo_field_tables(self._h, _p(E, _i64p), _p(L, _i64p), _p(Z, _i64p), ctypes.byref(ze))
        return E, L, Z, ze.value

    # ---- element-wise -------------------------------------------------------------------------------------
    def ufunc(self, op: int, a, b=None):
        """Broadcasting element-wise op; returns uint64 array.  Raises ZeroDivisionError like the reference."""
        a = _as_u64(a)
        if b is not None:
            if op == POW:
                b = np.ascontiguousarray(np.asarray(b, dtype=np.int64)).view(np.uint64)
            else:
                b = _as_u64(b)
            a, b = np.broadcast_arrays(a, b)
            a = np.ascontiguousarray(a)
            b = np.ascontiguousarray(b)
        out = np.empty(a.shape, dtype=np.uint64)
        rc = lib().gfo_ufunc(self._h, op, _p(a, _u64p), 1, _p(b, _u64p) if b is not None else None, 1, _p(out, _u64p),
                             a.size)
        if rc == ZERO_DIVISION:
            raise ZeroDivisionError("Cannot compute the multiplicative inverse of 0 in a Galois field.")
        if rc:
            raise ValueError(f"oracle error {rc}")
        return out

    def add(self, a, b): return self.ufunc(ADD, a, b)
    def sub(self, a, b): return self.ufunc(SUB, a, b)
    def mul(self, a, b): return self.ufunc(MUL, a, b)
    def div(self, a, b): return self.ufunc(DIV, a, b)
    def neg(self, a): return self.ufunc(NEG, a)
    def recip(self, a): return self.ufunc(RECIP, a)
    def pow(self, a, b): return self.ufunc(POW, a, b)

    def ufunc_u8(self, op: int, a: np.ndarray, b: np.ndarray | None = None) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint8)
        b = np.ascontiguousarray(b, dtype=np.uint8) if b is not None else a
        out = np.empty_like(a)
        rc = lib().gfo_ufunc_u8(self._h, op, _p(a, _u8p), _p(b, _u8p), _p(out, _u8p), a.size)
        if rc == ZERO_DIVISION:
            raise ZeroDivisionError("Cannot compute the multiplicative inverse of 0 in a Galois field.")
        if rc:
            raise ValueError(f"oracle error {rc}")
        return out

    # ---- transforms ---------------------------------------------------------------------------------------
    def root_of_unity(self, n: int) -> int:
        """FieldArray.primitive_root_of_unity _fields/_array.py:1182-1187: alpha ** ((q-1)/n)."""
        if (self.q - 1) % n != 0:
            raise ValueError(f"There are no primitive {n}-th roots of unity in GF({self.q}).")
        return int(self.pow([self.alpha], [(self.q - 1) // n])[0])

    def ntt(self, x, omega: int | None = None, inverse: bool = False, scaled: bool = True):
        """fft_jit.__call__ _domains/_function.py:177-212 (norm handling as galois.intt: scaled=True divides by n)."""
        x = _as_u64(x)
        n = x.size
        if omega is None:
            omega = self.root_of_unity(n)
        if inverse:
            omega = int(self.recip([omega])[0])
        facs = np.array(prime_factors(n), dtype=np.int64)
        out = np.empty(n, dtype=np.uint64)
        rc = lib().gfo_ntt(self._h, _p(x, _u64p), n, omega, _p(facs, _i64p), facs.size, _p(out, _u64p))
        if rc:
            raise ValueError(f"oracle ntt error {rc}")
        if inverse and scaled:
            out = self.div(out, [n % self.p])
        return out

    def ntt_u32_pow2(self, x: np.ndarray, omega: int) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.uint32)
        out = np.empty_like(x)
        rc = lib().gfo_ntt_u32_pow2(self.p, _p(x, _u32p), x.size, omega, _p(out, _u32p))
        if rc:
            raise ValueError(f"oracle ntt error {rc}")
        return out

    # ---- linear algebra / polys ---------------------------------------------------------------------------
    def matmul(self, A, B):
        A = _as_u64(A); B = _as_u64(B)
        M, K = A.shape; K2, N = B.shape
        assert K == K2
        C = np.empty((M, N), dtype=np.uint64)
        lib().gfo_matmul(self._h, _p(A, _u64p), _p(B, _u64p), _p(C, _u64p), M, K, N)
        return C

    def poly_eval(self, coeffs_desc, values):
        c = _as_u64(coeffs_desc); v = _as_u64(values)
        y = np.empty(v.size, dtype=np.uint64)
        lib().gfo_poly_eval(self._h, _p(c, _u64p), c.size, _p(v, _u64p), v.size, _p(y, _u64p))
        return y

    def convolve(self, a, b):
        a = _as_u64(a); b = _as_u64(b)
        c = np.empty(a.size + b.size - 1, dtype=np.uint64)
        lib().gfo_convolve(self._h, _p(a, _u64p), a.size, _p(b, _u64p), b.size, _p(c, _u64p))
        return c

    def berlekamp_massey(self, S):
        """Returns the connection polynomial in degree-descending order like berlekamp_massey_jit."""
        S = _as_u64(S)
        C = np.zeros(S.size, dtype=np.uint64)
        ln = ctypes.c_int64(0)
        lib().gfo_berlekamp_massey(self._h, _p(S, _u64p), S.size, _p(C, _u64p), ctypes.byref(ln))
        return C[: ln.value][::-1].copy()


def prime_factors(n: int) -> list[int]:
    """fft_jit._prime_factors _function.py:214-229: prime factors ascending with multiplicity."""
    out, d = [], 2
    while d * d <= n:
        while n % d == 0:
            out.append(d)
            n //= d
        d += 1
    if n > 1:
        out.append(n)
    return out


class OracleRS:
    """ReedSolomon(n, k) over an OracleField, systematic (ReedSolomon.__init__ _codes/_reed_solomon.py:111-218)."""

    def __init__(self, field: OracleField, n: int, k: int, alpha: int | None = None, c: int = 1):
        self.field, self.n, self.k, self.c = field, n, k, c
        self.d = n - k + 1
        self.alpha = field.root_of_unity(n) if alpha is None else int(alpha)
        roots = np.zeros(max(n - k, 1), dtype=np.uint64)
        g = np.zeros(n - k + 1, dtype=np.uint64)
        P = np.zeros((k, max(n - k, 1)), dtype=np.uint64)
        lib().gfo_rs_construct(field._h, n, k, self.alpha, c, _p(roots, _u64p), _p(g, _u64p), _p(P, _u64p))
        self.roots = roots[: n - k]
        self.generator_poly = g  # descending, monic
        self.P = P[:, : n - k]
        self.G = np.hstack([np.eye(k, dtype=np.uint64), self.P])
        # H = power.outer(roots, arange(n-1, -1, -1))  (_reed_solomon.py:218)
        self.H = np.stack([field.pow(np.full(n, r, dtype=np.uint64), np.arange(n - 1, -1, -1)) for r in self.roots]) \
            if n - k > 0 else np.zeros((0, n), dtype=np.uint64)

    def encode(self, message):
        """_LinearCode._encode_message _codes/_linear.py:270-284 (systematic, shortened allowed)."""
        m = _as_u64(message)
        one_d = m.ndim == 1
        m = np.atleast_2d(m)
        ks = m.shape[1]
        pad = self.k - ks
        parity = self.field.matmul(m, self.P[pad:, :])
        cw = np.hstack([m, parity])
        return cw[0] if one_d else cw

    def encode_u8(self, message: np.ndarray) -> np.ndarray:
        m = np.ascontiguousarray(np.atleast_2d(message), dtype=np.uint8)
        N, ks = m.shape
        nk = self.n - self.k
        P8 = np.ascontiguousarray(self.P, dtype=np.uint8)
        out = np.empty((N, ks + nk), dtype=np.uint8)
        rc = lib().gfo_rs_encode_u8(self.field._h, _p(m, _u8p), N, ks, self.k, nk, _p(P8, _u8p), _p(out, _u8p))
        assert rc == 0
        return out

    def detect(self, codeword):
        """_LinearCode._detect_errors _codes/_linear.py:286-298."""
        cw = np.atleast_2d(_as_u64(codeword))
        ns = cw.shape[1]
        syn = self.field.matmul(cw, np.ascontiguousarray(self.H[:, self.n - ns:].T))
        return ~np.all(syn == 0, axis=1)

    def decode(self, codeword, erasures=None):
        """bch_decode_jit via _LinearCode.decode.  Returns (dec_codeword (N, ns) uint64, n_errors (N,) int64)."""
        cw = np.ascontiguousarray(np.atleast_2d(_as_u64(codeword)))
        N, ns = cw.shape
        er = None
        if erasures is not None:
            er = np.ascontiguousarray(np.atleast_2d(np.asarray(erasures)).astype(np.uint8))
            assert er.shape == cw.shape
        dec = np.empty_like(cw)
        nerr = np.empty(N, dtype=np.int64)
        roots = np.ascontiguousarray(self.roots)
        rc = lib().gfo_rs_decode(self.field._h, _p(cw, _u64p), _p(er, _u8p) if er is not None else None, N, ns, self.n,
                                 self.alpha, self.c, _p(roots, _u64p), roots.size, _p(dec, _u64p), _p(nerr, _i64p))
        if rc:
            raise ValueError(f"oracle decode error {rc}")
        return dec, nerr

    def decode_u8(self, codeword: np.ndarray, erasures=None):
        cw = np.ascontiguousarray(np.atleast_2d(codeword), dtype=np.uint8)
        N, ns = cw.shape
        er = None
        if erasures is not None:
            er = np.ascontiguousarray(np.atleast_2d(np.asarray(erasures)).astype(np.uint8))
        dec = np.empty_like(cw)
        nerr = np.empty(N, dtype=np.int64)
        roots = np.ascontiguousarray(self.roots)
        rc = lib().gfo_rs_decode_u8(self.field._h, _p(cw, _u8p), _p(er, _u8p) if er is not None else None, N, ns, self.n,
                                    self.alpha, self.c, _p(roots, _u64p), roots.size, _p(dec, _u8p), _p(nerr, _i64p))
        if rc:
            raise ValueError(f"oracle decode error {rc}")
        return dec, nerr
