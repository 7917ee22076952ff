"""
Field linear algebra on device arrays: the host-side mirror of the reference's linear-algebra front ends
(paths relative to /root/reference/src/galois):

  * matmul_jit.__call__ ................ _domains/_linalg.py:190-281 (np.matmul 1-D promotion, batch broadcasting)
  * dot / vdot / inner / outer ......... _domains/_linalg.py:83-180 (prime fields follow NumPy's own shape rules through
                                          _lapack_linalg :21-75; other fields follow the explicit branches)
  * row_reduce / lu / plu .............. _domains/_linalg.py:315-424, FieldArray methods _fields/_array.py:1412-1540
  * det / matrix_rank / inv / solve .... _domains/_linalg.py:427-548
  * row/column/left-null/null space .... _fields/_array.py:1541-1760
The arithmetic is gfa_matmul / gfa_row_reduce / gfa_plu_decompose (include/galois_amd.h); this module only checks
arguments, lays operands out and reproduces the reference's exception types.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib as L
from ._array import FieldArray, _ptr, _stream


def _verify_same_field(a, b, what: str):
    if not isinstance(a, FieldArray) or not isinstance(b, FieldArray) or type(a) is not type(b):
        raise TypeError(f"Operation {what!r} requires both operands to be arrays over the same field, not {type(a)} and {type(b)}.")


def _matmul_3d(cls, ta: torch.Tensor, tb: torch.Tensor, a_stride: int, b_stride: int, batch: int, M: int, K: int, N: int,
               gfa_dtype: int) -> torch.Tensor:
    out = torch.empty((batch, M, N), dtype=ta.dtype, device=ta.device)
    if cls._limbed:  # order >= 2^64: ta / tb carry one complex128 entry (= two 64-bit limbs) per element
        L.check(L.lib().gfa_wide_matmul(cls._wide_handle, _ptr(ta), _ptr(tb), _ptr(out), batch, M, K, N, a_stride, b_stride, _stream()),
                "gfa_wide_matmul")
        return out
    L.check(L.lib().gfa_matmul(cls._handle, _ptr(ta), _ptr(tb), _ptr(out), batch, M, K, N, a_stride, b_stride, gfa_dtype,
                               _stream()), "gfa_matmul")
    return out


def matmul(A: FieldArray, B: FieldArray) -> FieldArray:
    """np.matmul / the @ operator (matmul_jit.__call__, _linalg.py:190-281)."""
    _verify_same_field(A, B, "matmul")
    cls = type(A)
    if not (A.ndim >= 1 and B.ndim >= 1):
        raise ValueError(f"Operation 'matmul' requires both arrays have dimension at least 1, not {A.ndim}-D and {B.ndim}-D.")
    ta, tb = (A._af_tens(A), A._af_tens(B)) if cls._limbed else (A._t, A._same_storage(B))
    a_vec, b_vec = ta.dim() == 1, tb.dim() == 1
    if a_vec:
        ta = ta.reshape(1, -1)
    if b_vec:
        tb = tb.reshape(-1, 1)
    M, K = ta.shape[-2], ta.shape[-1]
    K2, N = tb.shape[-2], tb.shape[-1]
    if K != K2:
        raise ValueError(
            f"Operation 'matmul' requires the last dimension of 'A' to match the second-to-last dimension of 'B', "
            f"not {tuple(A.shape)} and {tuple(B.shape)}."
        )
    batch_shape = tuple(torch.broadcast_shapes(ta.shape[:-2], tb.shape[:-2]))
    batch = int(math.prod(batch_shape)) if batch_shape else 1

    def lay_out(t, rows, cols):
        if math.prod(t.shape[:-2]) == 1:  # one matrix broadcast over the batch: stride 0, no copy
            return t.reshape(rows, cols).contiguous(), 0
        return t.expand(batch_shape + (rows, cols)).contiguous().reshape(batch, rows, cols), rows * cols

    ta3, sa = lay_out(ta, M, K)
    tb3, sb = lay_out(tb, K, N)
    out = _matmul_3d(cls, ta3, tb3, sa, sb, batch, M, K, N, None if cls._limbed else A._gfa_dtype())
    if a_vec and b_vec:
        final = batch_shape
    elif a_vec:
        final = batch_shape + (N,)
    elif b_vec:
        final = batch_shape + (M,)
    else:
        final = batch_shape + (M, N)
    if cls._limbed:
        return A._af_wrap(out.reshape(final))
    return cls._wrap(out.reshape(final), A._np_dtype)


def dot(a: FieldArray, b: FieldArray) -> FieldArray:
    """np.dot (dot_jit.__call__, _linalg.py:83-113)."""
    _verify_same_field(a, b, "dot")
    cls = type(a)
    if a.ndim == 0 or b.ndim == 0:
        return a * b
    if a.ndim == 1 and b.ndim == 1:
        if a.shape != b.shape:
            raise ValueError(f"shapes {tuple(a.shape)} and {tuple(b.shape)} not aligned")
        return matmul(a, b)
    if a.ndim == 2 and b.ndim == 2:
        return matmul(a, b)
    if a.ndim >= 2 and b.ndim == 1:
        return matmul(a, b)
    if not cls.is_prime_field or cls._limbed:
        raise NotImplementedError(
            "Currently 'dot' is only supported up to 2-D matrices. "
            "Please open a GitHub issue at https://github.com/mhostetter/galois/issues."
        )
    # prime fields follow np.dot itself (_lapack_linalg): sum over the last axis of a and the second-to-last of b
    if b.ndim == 1:
        return matmul(a, b)
    K = a.shape[-1]
    if b.shape[-2] != K:
        raise ValueError(f"shapes {tuple(a.shape)} and {tuple(b.shape)} not aligned")
    a2 = a.reshape(-1, K) if a.ndim > 1 else a.reshape(1, K)
    tb = a._same_storage(b)
    b2 = cls._wrap(tb.movedim(-2, 0).reshape(K, -1).contiguous(), b._np_dtype)
    res = matmul(a2, b2)
    shape = tuple(a.shape[:-1]) + tuple(b.shape[:-2]) + (b.shape[-1],)
    return res.reshape(shape)


def vdot(a: FieldArray, b: FieldArray) -> FieldArray:
    """np.vdot (vdot_jit.__call__, _linalg.py:116-132): flattened dot product, no conjugation in a finite field."""
    _verify_same_field(a, b, "vdot")
    fa, fb = a.flatten(), b.flatten()
    if fa.size != fb.size:
        raise ValueError(f"cannot reshape array of size {fb.size} into shape {tuple(fa.shape)}")
    return matmul(fa, fb)


def inner(a: FieldArray, b: FieldArray) -> FieldArray:
    """np.inner (inner_jit.__call__, _linalg.py:135-156)."""
    _verify_same_field(a, b, "inner")
    cls = type(a)
    if a.ndim == 0 or b.ndim == 0:
        return a * b
    if not a.shape[-1] == b.shape[-1]:
        raise ValueError(
            f"Operation 'inner' requires 'a' and 'b' to have the same last dimension, not {tuple(a.shape)} and {tuple(b.shape)}."
        )
    if cls.is_prime_field and not cls._limbed:
        # np.inner: out[i..., j...] = sum_k a[i..., k] b[j..., k]
        K = a.shape[-1]
        a2 = a.reshape(-1, K)
        b2 = cls._wrap(a._same_storage(b).reshape(-1, K).t().contiguous(), b._np_dtype)
        return matmul(a2, b2).reshape(tuple(a.shape[:-1]) + tuple(b.shape[:-1]))
    return np.add.reduce(a * b, axis=-1)


def outer(a: FieldArray, b: FieldArray) -> FieldArray:
    """np.outer (outer_jit.__call__, _linalg.py:159-173)."""
    _verify_same_field(a, b, "outer")
    return np.multiply.outer(a.flatten(), b.flatten())


# ---- elimination-based routines --------------------------------------------------------------------------------------
# Tensors below carry ONE entry per field element: the storage tensor itself, or -- for the two-limb fields of order >= 2^64 --
# its complex128 view (a bit container for the limb pair: moved, never computed on; galois_amd/_wide.py).
def _elems(A: FieldArray) -> torch.Tensor:
    return A._af_tens(A) if type(A)._limbed else A._t


def _rewrap(A: FieldArray, t: torch.Tensor) -> FieldArray:
    return A._af_wrap(t) if type(A)._limbed else type(A)._wrap(t, A._np_dtype)


def _dtype_code(A: FieldArray):
    return None if type(A)._limbed else A._gfa_dtype()


def _row_reduce_t(cls, t: torch.Tensor, ncols: int, gfa_dtype):
    """In-place Gauss-Jordan on a contiguous (batch, m, n) tensor; returns the per-matrix pivot counts (device int64)."""
    batch, m, n = t.shape
    ranks = torch.zeros(batch, dtype=torch.int64, device=t.device)
    if batch and m and n:
        if cls._limbed:
            L.check(L.lib().gfa_wide_row_reduce(cls._wide_handle, _ptr(t), batch, m, n, ncols, _ptr(ranks), _stream()), "gfa_wide_row_reduce")
        else:
            L.check(L.lib().gfa_row_reduce(cls._handle, _ptr(t), batch, m, n, ncols, _ptr(ranks), gfa_dtype, _stream()),
                    "gfa_row_reduce")
    return ranks


def row_reduce(A: FieldArray, ncols: int | None = None, eye: str = "left") -> FieldArray:
    """FieldArray.row_reduce (_fields/_array.py:1412-1468) over row_reduce_jit (_linalg.py:315-351)."""
    if eye not in ["left", "right"]:
        raise ValueError(f"Argument 'eye' must be one of ['left', 'right'], not {eye!r}.")
    if not A.ndim == 2:
        raise ValueError(f"Only 2-D matrices can be converted to reduced row echelon form, not {A.ndim}-D.")
    cls = type(A)
    t = _elems(A)
    if eye == "right":
        t = torch.flip(t, dims=(0, 1))
    t = t.contiguous().clone().reshape(1, *t.shape)
    ncols = t.shape[2] if ncols is None else int(ncols)
    _row_reduce_t(cls, t, ncols, _dtype_code(A))
    t = t[0]
    if eye == "right":
        t = torch.flip(t, dims=(0, 1)).contiguous()
    return _rewrap(A, t)


def _row_reduce_with_rank(A: FieldArray, ncols: int | None = None):
    cls = type(A)
    t0 = _elems(A)
    t = t0.contiguous().clone().reshape(1, *t0.shape)
    ranks = _row_reduce_t(cls, t, t.shape[2] if ncols is None else int(ncols), _dtype_code(A))
    return _rewrap(A, t[0]), int(ranks[0].item())


def row_reduce_batched(A: FieldArray, ncols: int | None = None):
    """Device extension: reduced row echelon form of every matrix of a (batch, m, n) stack in one launch.  Returns
    (stack of RREFs, ranks as a host int64 array)."""
    if not A.ndim == 3:
        raise ValueError(f"row_reduce_batched expects a 3-D stack of matrices, not {A.ndim}-D.")
    cls = type(A)
    t = _elems(A).contiguous().clone()
    ranks = _row_reduce_t(cls, t, t.shape[2] if ncols is None else int(ncols), _dtype_code(A))
    return _rewrap(A, t), ranks.cpu().numpy()


def _plu(A: FieldArray, pivoting: bool, want_l: bool = True, want_p: bool = True, want_det: bool = False):
    cls = type(A)
    t = _elems(A).contiguous().clone()
    batched = t.dim() == 3
    if not batched:
        t = t.reshape(1, *t.shape)
    batch, m, n = t.shape
    lo = torch.empty((batch, m, m), dtype=t.dtype, device=t.device) if want_l else None
    po = torch.empty((batch, m, m), dtype=t.dtype, device=t.device) if want_p else None
    nperm = torch.zeros(batch, dtype=torch.int64, device=t.device)
    det = torch.empty(batch, dtype=t.dtype, device=t.device) if want_det else None
    err = torch.zeros(1, dtype=torch.int32, device=t.device)
    if cls._limbed:
        L.check(L.lib().gfa_wide_plu_decompose(cls._wide_handle, _ptr(t), _ptr(lo) if want_l else None, _ptr(po) if want_p else None, batch,
                                               m, n, 1 if pivoting else 0, _ptr(nperm), _ptr(det) if want_det else None, _stream(), _ptr(err)),
                "gfa_wide_plu_decompose")
    else:
        L.check(L.lib().gfa_plu_decompose(cls._handle, _ptr(t), _ptr(lo) if want_l else None, _ptr(po) if want_p else None, batch,
                                          m, n, 1 if pivoting else 0, _ptr(nperm), _ptr(det) if want_det else None,
                                          A._gfa_dtype(), _stream(), _ptr(err)), "gfa_plu_decompose")
    if not pivoting and int(err.item()) & L.DEVERR_NO_LU:
        raise ValueError("The LU decomposition of 'A' does not exist. Use the PLU decomposition instead.")
    return t, lo, po, nperm, det, batched


def lu_decompose(A: FieldArray):
    """FieldArray.lu_decompose (_fields/_array.py:1471-1501) over lu_decompose_jit (_linalg.py:354-384)."""
    if not A.ndim == 2:
        raise ValueError(f"Argument 'A' must be a 2-D matrix, not have shape {tuple(A.shape)}.")
    m, n = A.shape
    if m - 1 > n:
        raise IndexError(f"index {n} is out of bounds for axis 1 with size {n}")  # what Ai[i, i] raises in the reference
    u, lo, _, _, _, _ = _plu(A, pivoting=False, want_p=False)
    return _rewrap(A, lo[0]), _rewrap(A, u[0])


def plu_decompose(A: FieldArray):
    """FieldArray.plu_decompose (_fields/_array.py:1504-1538) over plu_decompose_jit (_linalg.py:387-424)."""
    if not A.ndim == 2:
        raise ValueError(f"Argument 'A' must be a 2-D matrix, not have shape {tuple(A.shape)}.")
    u, lo, po, _, _, _ = _plu(A, pivoting=True)
    return (_rewrap(A, po[0].t().contiguous()), _rewrap(A, lo[0]), _rewrap(A, u[0]))


def det(A: FieldArray) -> FieldArray:
    """np.linalg.det (det_jit.__call__, _linalg.py:447-477)."""
    if not (A.ndim == 2 and A.shape[0] == A.shape[1]):
        raise np.linalg.LinAlgError(f"Argument 'A' must be square, not {tuple(A.shape)}.")
    cls = type(A)
    if A.shape[0] == 0:
        return cls.Ones(()) if cls._limbed else cls._wrap(torch.ones((), dtype=A._t.dtype, device=A._t.device), A._np_dtype)
    _, _, _, _, d, _ = _plu(A, pivoting=True, want_l=False, want_p=False, want_det=True)
    return _rewrap(A, d[0])


def det_batched(A: FieldArray) -> FieldArray:
    """Device extension: determinants of a (batch, n, n) stack in one launch."""
    if not (A.ndim == 3 and A.shape[1] == A.shape[2]):
        raise np.linalg.LinAlgError(f"Argument 'A' must be a stack of square matrices, not {tuple(A.shape)}.")
    _, _, _, _, d, _ = _plu(A, pivoting=True, want_l=False, want_p=False, want_det=True)
    return _rewrap(A, d)


def matrix_rank(A: FieldArray) -> int:
    """np.linalg.matrix_rank (matrix_rank_jit.__call__, _linalg.py:485-492)."""
    if not A.ndim == 2:
        raise ValueError(f"Only 2-D matrices can be converted to reduced row echelon form, not {A.ndim}-D.")
    return _row_reduce_with_rank(A)[1]


def _augment_identity(A: FieldArray) -> torch.Tensor:
    t = _elems(A)
    if type(A)._limbed:  # the element 1 is the limb pair (1, 0): built as limbs, then viewed like the rest
        n = t.shape[-2]
        limbs = torch.zeros((n, n, 2), dtype=torch.int64, device=t.device)
        limbs[torch.arange(n), torch.arange(n), 0] = 1
        eye = limbs.view(torch.complex128).squeeze(-1)
    else:
        eye = torch.eye(t.shape[-2], dtype=t.dtype, device=t.device)
    if t.dim() == 3:
        eye = eye.expand(t.shape[0], -1, -1)
    return torch.cat([t, eye], dim=-1).contiguous()


def inv(A: FieldArray) -> FieldArray:
    """np.linalg.inv (inv_jit.__call__, _linalg.py:495-520): row reduction of [A | I] over the first n columns."""
    if not (A.ndim == 2 and A.shape[0] == A.shape[1]):
        raise np.linalg.LinAlgError(f"Argument 'A' must be square, not {tuple(A.shape)}.")
    cls = type(A)
    n = A.shape[0]
    ai = _augment_identity(A).reshape(1, n, 2 * n)
    ranks = _row_reduce_t(cls, ai, n, _dtype_code(A))
    rank = int(ranks[0].item())
    if not rank == n:
        raise np.linalg.LinAlgError(
            f"Argument 'A' is singular and not invertible because it does not have full rank of {n}, but rank of {rank}."
        )
    return _rewrap(A, ai[0, :, n:].contiguous())


def inv_batched(A: FieldArray) -> FieldArray:
    """Device extension: inverses of a (batch, n, n) stack in one launch; LinAlgError if any matrix is singular."""
    if not (A.ndim == 3 and A.shape[1] == A.shape[2]):
        raise np.linalg.LinAlgError(f"Argument 'A' must be a stack of square matrices, not {tuple(A.shape)}.")
    cls = type(A)
    n = A.shape[1]
    ai = _augment_identity(A)
    ranks = _row_reduce_t(cls, ai, n, _dtype_code(A))
    if not bool((ranks == n).all()):
        bad = int((ranks != n).nonzero()[0].item())
        raise np.linalg.LinAlgError(f"Matrix {bad} of the stack is singular and not invertible.")
    return _rewrap(A, ai[:, :, n:].contiguous())


def solve(A: FieldArray, b: FieldArray) -> FieldArray:
    """np.linalg.solve (solve_jit.__call__, _linalg.py:523-548)."""
    _verify_same_field(A, b, "solve")
    if not (A.ndim == 2 and A.shape[0] == A.shape[1]):
        raise np.linalg.LinAlgError(f"Argument 'A' must be square, not {tuple(A.shape)}.")
    if b.ndim not in [1, 2]:
        raise np.linalg.LinAlgError(f"Argument 'b' must have dimension equal to 'A' or one less, not {b.ndim}.")
    if not A.shape[-1] == b.shape[0]:
        raise np.linalg.LinAlgError(
            f"The last dimension of 'A' must equal the first dimension of 'b', not {tuple(A.shape)} and {tuple(b.shape)}."
        )
    return matmul(inv(A), b)


def row_space(A: FieldArray) -> FieldArray:
    """FieldArray.row_space (_fields/_array.py:1541-1590)."""
    if not A.ndim == 2:
        raise ValueError(f"Only 2-D matrices have a row space, not {A.ndim}-D.")
    rre, rank = _row_reduce_with_rank(A)
    return rre[0:rank, :]


def column_space(A: FieldArray) -> FieldArray:
    if not A.ndim == 2:
        raise ValueError(f"Only 2-D matrices have a column space, not {A.ndim}-D.")
    return row_space(A.T)


def left_null_space(A: FieldArray) -> FieldArray:
    """FieldArray.left_null_space (_fields/_array.py:1639-1703)."""
    if not A.ndim == 2:
        raise ValueError(f"Only 2-D matrices have a left null space, not {A.ndim}-D.")
    cls = type(A)
    m, n = A.shape
    ai = _augment_identity(A).reshape(1, m, n + m)
    p = int(_row_reduce_t(cls, ai, n, _dtype_code(A))[0].item())
    ln = _rewrap(A, ai[0, p:, n:].contiguous())
    if ln.shape[0] == 0:
        return ln
    return row_reduce(ln)


def null_space(A: FieldArray) -> FieldArray:
    if not A.ndim == 2:
        raise ValueError(f"Only 2-D matrices have a null space, not {A.ndim}-D.")
    return left_null_space(A.T)
