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
        L.gfo_bch_decode.argtypes = [ctypes.c_void_p, ctypes.c_uint64, _u64p, _u8p, ctypes.c_int64, ctypes.c_int64,
                                     ctypes.c_int64, ctypes.c_uint64, ctypes.c_int64, _u64p, ctypes.c_int64, _u64p, _i64p]
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

    # matrix routines of _domains/_linalg.py, restated with the same pivot rules on uint64 host arrays
    def _outer_sub(self, A, rows, col_factors, pivot_row):
        """A[rows, :] -= outer(col_factors, pivot_row)."""
        if len(rows) == 0:
            return
        prod = self.mul(np.repeat(col_factors[:, None], pivot_row.size, axis=1), np.tile(pivot_row, (len(rows), 1)))
        A[rows, :] = self.sub(A[rows, :], prod)

    def row_reduce(self, A, ncols=None):
        """row_reduce_jit.__call__ _linalg.py:315-351.  Returns (A_rre, p)."""
        A = _as_u64(A).copy()
        m, n = A.shape
        ncols = n if ncols is None else ncols
        p = 0
        if m == 0:
            return A, 0
        for j in range(ncols):
            idxs = np.nonzero(A[p:, j])[0]
            if idxs.size == 0:
                continue
            i = p + int(idxs[0])
            A[[p, i], :] = A[[i, p], :]
            A[p, :] = self.div(A[p, :], np.full(n, A[p, j], dtype=np.uint64))
            idxs = [int(r) for r in np.nonzero(A[:, j])[0] if r != p]
            self._outer_sub(A, idxs, A[idxs, j].copy(), A[p, :].copy())
            p += 1
            if p == m:
                break
        return A, p

    def lu_decompose(self, A):
        """lu_decompose_jit.__call__ _linalg.py:354-384.  Returns (L, U); ValueError if a row exchange is needed."""
        Ai = _as_u64(A).copy()
        m = Ai.shape[0]
        L = np.eye(m, dtype=np.uint64)
        for i in range(0, m - 1):
            if Ai[i, i] == 0:
                idxs = np.nonzero(Ai[i:, i])[0]
                if idxs.size == 0:
                    L[i, i] = 1
                    continue
                raise ValueError("The LU decomposition of 'A' does not exist. Use the PLU decomposition instead.")
            l = self.div(Ai[i + 1:, i], np.full(m - i - 1, Ai[i, i], dtype=np.uint64))
            self._outer_sub(Ai, list(range(i + 1, m)), l, Ai[i, :].copy())
            L[i + 1:, i] = l
        return L, Ai

    def plu_decompose(self, A):
        """plu_decompose_jit.__call__ _linalg.py:387-424.  Returns (P (column permutation = P_row.T), L, U, N_perm)."""
        Ai = _as_u64(A).copy()
        m, n = Ai.shape
        L = np.zeros((m, m), dtype=np.uint64)
        P = np.eye(m, dtype=np.uint64)
        nperm = 0
        for i in range(0, min(m, n)):
            if Ai[i, i] == 0:
                idxs = np.nonzero(Ai[i:, i])[0]
                if idxs.size == 0:
                    L[i, i] = 1
                    continue
                j = i + int(idxs[0])
                P[[i, j], :] = P[[j, i], :]
                Ai[[i, j], :] = Ai[[j, i], :]
                L[[i, j], :] = L[[j, i], :]
                nperm += 1
            l = self.div(Ai[i + 1:, i], np.full(m - i - 1, Ai[i, i], dtype=np.uint64)) if i + 1 < m else np.zeros(0, np.uint64)
            self._outer_sub(Ai, list(range(i + 1, m)), l, Ai[i, :].copy())
            L[i, i] = 1
            L[i + 1:, i] = l
        L[-1, -1] = 1
        return P.T.copy(), L, Ai, nperm

    def det(self, A):
        """det_jit.__call__ _linalg.py:447-477 (the 2x2 / 3x3 closed forms give the same field element)."""
        A = _as_u64(A)
        assert A.ndim == 2 and A.shape[0] == A.shape[1]
        _, L, U, nperm = self.plu_decompose(A)
        d = 1
        for i in range(A.shape[0]):
            d = int(self.mul([d], [int(L[i, i])])[0])
            d = int(self.mul([d], [int(U[i, i])])[0])
        if nperm % 2:
            d = int(self.neg([d])[0])
        return d

    def matrix_rank(self, A):
        A_rre, _ = self.row_reduce(A)
        return int(np.sum(~np.all(A_rre == 0, axis=1)))

    def inv(self, A):
        """inv_jit.__call__ _linalg.py:495-520."""
        A = _as_u64(A)
        n = A.shape[0]
        assert A.ndim == 2 and A.shape[1] == n
        AI = np.concatenate([A, np.eye(n, dtype=np.uint64)], axis=-1)
        AI_rre, _ = self.row_reduce(AI, ncols=n)
        rank = int(np.sum(~np.all(AI_rre[:, 0:n] == 0, axis=1)))
        if rank != n:
            raise np.linalg.LinAlgError(f"Argument 'A' is singular and not invertible because it does not have full rank of {n}, but rank of {rank}.")
        return AI_rre[:, -n:].copy()

    def solve(self, A, b):
        """solve_jit.__call__ _linalg.py:523-548."""
        b = _as_u64(b)
        Ainv = self.inv(A)
        if b.ndim == 1:
            return self.matmul(Ainv, b.reshape(-1, 1)).reshape(-1)
        return self.matmul(Ainv, b)

    def row_space(self, A):
        """FieldArray.row_space _fields/_array.py:1541-1590."""
        A_rre, _ = self.row_reduce(A)
        rank = int(np.sum(~np.all(A_rre == 0, axis=1)))
        return A_rre[0:rank, :]

    def column_space(self, A):
        return self.row_space(_as_u64(A).T)

    def left_null_space(self, A):
        """FieldArray.left_null_space _fields/_array.py:1639-1703."""
        A = _as_u64(A)
        m, n = A.shape
        AI = np.concatenate([A, np.eye(m, dtype=np.uint64)], axis=-1)
        AI_rre, p = self.row_reduce(AI, ncols=n)
        LN = AI_rre[p:, n:]
        LN, _ = self.row_reduce(LN)
        return LN

    def null_space(self, A):
        return self.left_null_space(_as_u64(A).T)

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


# ---- polynomials over the prime field GF(p), coefficients highest degree first, plain Python ints ----------------
def _pp_trim(a):
    a = list(a)
    while len(a) > 1 and a[0] == 0:
        a.pop(0)
    return a


def _pp_mul(a, b, p):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % p
    return _pp_trim(out)


def _pp_divmod(a, b, p):
    a, b = _pp_trim(a), _pp_trim(b)
    if len(a) < len(b):
        return [0], a
    inv = pow(b[0], -1, p)
    a = list(a)
    q = []
    for i in range(len(a) - len(b) + 1):
        c = a[i] * inv % p
        q.append(c)
        for j, y in enumerate(b):
            a[i + j] = (a[i + j] - c * y) % p
    return _pp_trim(q), _pp_trim(a[len(a) - len(b) + 1:] or [0])


class OracleBCH:
    """BCH(n, k) over the prime field GF(p) with syndromes in `ext` = GF(p^m)  (BCH.__init__ _codes/_bch.py:106-240,
    _generator_poly_from_d / _from_k :1178-1252, _CyclicCode.__init__ _codes/_cyclic.py:26-54)."""

    def __init__(self, ext: OracleField, n: int, k: int | None = None, d: int | None = None, alpha: int | None = None,
                 c: int = 1, systematic: bool = True):
        self.ext, self.p, self.n, self.c, self.systematic = ext, ext.p, n, c, systematic
        self.alpha = ext.root_of_unity(n) if alpha is None else int(alpha)
        if d is not None:
            g, roots = self._gen_from_d(d)
            kk = n - (len(g) - 1)
            if k not in (None, kk):
                raise ValueError(f"The requested [{n}, {k}, {d}] code is not consistent.")
            k = kk
        else:
            g, roots = self._gen_from_k(k)
            d = len(roots) + 1
        self.k, self.d = k, d
        self.generator_poly = g
        self.roots = np.array(roots, dtype=np.uint64)
        xn1 = [1] + [0] * (n - 1) + [self.p - 1]
        h, rem = _pp_divmod(xn1, g, self.p)
        assert rem == [0]
        self.parity_check_poly = h
        self.G = self._poly_to_generator_matrix(g, systematic)
        self.H = self._poly_to_generator_matrix(h[::-1], False)
        self._base = OracleField(self.p, 1, None, _smallest_primitive_root(self.p))

    # minimal polynomial over GF(p) of an element of GF(p^m): product over its distinct Frobenius conjugates
    # (FieldArray.minimal_poly, _fields/_array.py:1979-2050)
    def _minimal_poly(self, beta: int):
        E = self.ext
        conj, x = [], int(beta)
        while x not in conj:
            conj.append(x)
            x = int(E.pow(np.array([x], dtype=np.uint64), np.array([self.p]))[0])
        poly = [1]
        for r in conj:  # poly *= (x - r) in GF(p^m)
            nr = int(E.ufunc(SUB, np.array([0], dtype=np.uint64), np.array([r], dtype=np.uint64))[0])
            nxt = poly + [0]
            for i, cfe in enumerate(poly):
                t = int(E.mul(np.array([cfe], dtype=np.uint64), np.array([nr], dtype=np.uint64))[0])
                nxt[i + 1] = int(E.ufunc(ADD, np.array([nxt[i + 1]], dtype=np.uint64), np.array([t], dtype=np.uint64))[0])
            poly = nxt
        assert all(cf < self.p for cf in poly)
        return poly

    def _gen_from_d(self, d: int):
        E = self.ext
        roots = [int(E.pow(np.array([self.alpha], dtype=np.uint64), np.array([self.c + i]))[0]) for i in range(d - 1)]
        g, seen = [1], []
        for r in roots:
            mi = self._minimal_poly(r)
            if mi not in seen:
                g = _pp_mul(g, mi, self.p)
                seen.append(mi)
        return g, roots

    def _gen_from_k(self, k: int):
        n, m = self.n, self.ext.m
        possible_d = list(range((n - k) // m + 1, (n - k) + 2))
        found = False
        while possible_d:
            idx = len(possible_d) // 2
            d = possible_d[idx]
            g, roots = self._gen_from_d(d)
            if len(g) - 1 < n - k:
                possible_d = possible_d[idx + 1:]
            elif len(g) - 1 == n - k:
                found = True
                break
            else:
                possible_d = possible_d[:idx]
        if not found:
            raise ValueError(f"The BCH({n}, {k}) code does not exist.")
        best = (g, roots)
        while True:
            d += 1
            g, roots = self._gen_from_d(d)
            if len(g) - 1 == n - k:
                best = (g, roots)
            elif len(g) - 1 > n - k:
                break
        return best

    def _poly_to_generator_matrix(self, g, systematic: bool):
        """_codes/_cyclic.py:198-226 over GF(p)."""
        p, n = self.p, self.n
        deg = len(g) - 1
        k = n - deg
        if systematic:
            P = np.zeros((k, n - k), dtype=np.int64)
            if n - k > 0:
                inv = pow(g[-1], -1, p)
                P[0, :] = [(cf * inv) % p for cf in g[:-1]]
                for i in range(1, k):
                    P[i, 0] = 0
                    P[i, 1:] = P[i - 1, :-1]
                    if P[i - 1, -1] > 0:
                        P[i, :] = (P[i, :] - P[i - 1, -1] * P[0, :]) % p
            return np.hstack([np.eye(k, dtype=np.int64), P]).astype(np.uint64)
        G = np.zeros((k, n), dtype=np.int64)
        for i in range(k):
            G[i, i:i + deg + 1] = g
        return G.astype(np.uint64)

    def encode(self, message):
        """_LinearCode._encode_message _codes/_linear.py:270-284."""
        m = np.atleast_2d(_as_u64(message))
        pad = self.k - m.shape[1]
        if self.systematic:
            parity = self._base.matmul(m, np.ascontiguousarray(self.G[pad:, self.k:])) if self.n > self.k else m[:, :0]
            return np.hstack([m, parity])
        return self._base.matmul(m, np.ascontiguousarray(self.G[pad:, pad:]))

    def detect(self, codeword):
        cw = np.atleast_2d(_as_u64(codeword))
        ns = cw.shape[1]
        if self.n == self.k:
            return np.zeros(cw.shape[0], dtype=bool)
        syn = self._base.matmul(cw, np.ascontiguousarray(self.H[:, self.n - ns:].T))
        return ~np.all(syn == 0, axis=1)

    def decode(self, codeword, erasures=None):
        """bch_decode_jit (_codes/_bch.py:1337-1578).  Returns (dec_codeword (N, ns) int64, n_errors (N,) int64)."""
        cw = np.ascontiguousarray(np.atleast_2d(_as_u64(codeword)))
        N, ns = cw.shape
        er = None
        if erasures is not None:
            er = np.ascontiguousarray(np.atleast_2d(np.asarray(erasures)).astype(np.uint8))
        dec = np.empty_like(cw)
        nerr = np.empty(N, dtype=np.int64)
        roots = np.ascontiguousarray(self.roots) if self.roots.size else np.zeros(1, dtype=np.uint64)
        rc = lib().gfo_bch_decode(self.ext._h, self.p, _p(cw, _u64p), _p(er, _u8p) if er is not None else None, N, ns, self.n,
                                  self.alpha, self.c, _p(roots, _u64p), self.roots.size, _p(dec, _u64p), _p(nerr, _i64p))
        if rc:
            raise ValueError(f"oracle decode error {rc}")
        return dec.view(np.int64), nerr

    def message_of(self, dec_codeword):
        """_CyclicCode._convert_codeword_to_message (_codes/_cyclic.py:129-138)."""
        cw = np.atleast_2d(np.asarray(dec_codeword))
        ns = cw.shape[1]
        ks = self.k - (self.n - ns)
        if self.systematic:
            return cw[:, :ks]
        out = np.zeros((cw.shape[0], ks), dtype=np.int64)
        for i, row in enumerate(cw):
            q, _ = _pp_divmod([int(v) for v in row], self.generator_poly, self.p)
            q = [0] * (ks - len(q)) + q
            out[i] = q
        return out


def _smallest_primitive_root(p: int) -> int:
    if p == 2:
        return 1
    phi, fac, x, f = p - 1, [], p - 1, 2
    while f * f <= x:
        if x % f == 0:
            fac.append(f)
            while x % f == 0:
                x //= f
        f += 1
    if x > 1:
        fac.append(x)
    for g in range(2, p):
        if all(pow(g, phi // q, p) != 1 for q in fac):
            return g
    raise ValueError(p)
