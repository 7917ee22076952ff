"""
Field arrays over GF(q), q > 2^128 -- the k-limb path (r05).  The reference has no upper bound on the order: every field
beyond int64 is a dtype=object array of Python integers run through the pure-Python ufuncs (src/galois/_domains/_meta.py:38-41,
_fields/_ufunc.py:36-48).  Here the elements live on the GPU as 4, 8 or 16 little-endian 64-bit limbs (an int64 tensor with a
trailing limb axis; up to 1024 bits) and the element-wise ufunc surface runs in the kernels of csrc/gfa_big.hip: add, subtract,
multiply, divide, negative, reciprocal, power (arbitrary-size integer exponents), field * integer, square, ==, indexing and
reshaping.  ufunc.reduce / accumulate (np.sum, prod, cumsum, cumprod) are left folds of those kernels along the axis.  Exact
and simple, not fast: a coverage path, like the two-limb fields of galois_amd/_wide.py whose array surface this class inherits.
`where=` on calls, np.convolve and @ are compositions of those kernels.  What the two-limb fields have and these do not
(NotImplementedError): `where=` on reductions, row reduction, np.log, np.sqrt, np.fft, polynomial evaluation, and the NumPy data-movement
functions beyond reshape / transpose / concatenate / stack / where / copy.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from ._array import FieldArray
from ._wide import WideFieldArray

_M64 = (1 << 64) - 1


def limbs_for(order: int) -> int:
    bits = (order - 1).bit_length()
    for nl in (4, 8, 16):
        if bits <= 64 * nl:
            return nl
    raise NotImplementedError(
        f"A field of order 2^{bits} needs more than 16 limbs of 64 bits: the k-limb device representation of galois_amd stops at 1024 bits."
    )


def big_params(p: int, m: int, irr_int: int, nl: int) -> tuple[int, list[int]]:
    """(kind, 113 parameter words) for gfa_bfield_create (layout: include/galois_amd.h)."""
    q = p**m
    w = [0] * 113

    def put(at, value):
        for k in range(16):
            w[at + k] = (value >> (64 * k)) & _M64

    if m == 1:
        kind = 1
        put(0, p)
        w[16] = (-pow(p, -1, 1 << 64)) & _M64
        put(17, pow(2, 128 * nl, p))
        put(33, p - 2)
    elif p == 2:
        kind = 2
        w[0] = 2
        put(33, q - 2)
        put(49, irr_int ^ (1 << m))
    else:
        if p >= 2**32 or m > 32:
            raise NotImplementedError(f"GF({p}^{m}): extension fields of order > 2^128 need p < 2^32 and degree <= 32.")
        kind = 3
        w[0] = p
        put(65, (q - 1) // (p - 1) - 1)
        digits = []
        v = irr_int
        while v:
            digits.append(v % p)
            v //= p
        digits = digits[::-1]  # degree m .. 0
        for i, c in enumerate(digits[1:]):
            w[81 + i] = c
    return kind, w


class BigFieldArray(WideFieldArray):
    _NL = 4
    _kernels = ("gfa_big_binary", "gfa_big_unary", "gfa_big_power")

    def _no(self, what: str):
        raise NotImplementedError(f"{what} is not implemented for {type(self).name} (order > 2^128): element-wise ufuncs and their folds only.")

    # ---- folds: ufunc.reduce / accumulate as left folds of the element-wise kernel along the axis ----
    def _walk(self, op, axis):
        if self.ndim == 0:
            raise TypeError("cannot reduce on a scalar")
        t = self._t.reshape(-1, type(self)._NL) if axis is None else self._t.movedim(axis % self.ndim, 0)
        n = t.shape[0]
        if n == 0:
            raise ValueError("zero-size array to reduction operation which has no identity")
        cls = type(self)
        acc = cls._wrap(t[0].contiguous())
        yield acc
        for i in range(1, n):
            acc = self._binary(op, acc, cls._wrap(t[i].contiguous()))
            yield acc

    def _reduce(self, op, axis, keepdims: bool):
        cls = type(self)
        last = None
        for last in self._walk(op, axis):
            pass
        if keepdims:
            shape = (1,) * self.ndim if axis is None else tuple(1 if i == axis % self.ndim else d for i, d in enumerate(self.shape))
            return last.reshape(shape)
        return last

    def _accumulate(self, op, axis):
        cls = type(self)
        steps = [a._t for a in self._walk(op, axis)]
        return cls._wrap(torch.stack(steps, dim=0).movedim(0, axis % self.ndim).contiguous())

    def _reduce_kw(self, ufunc, op, axis, keepdims, where, initial):
        if self._kw_given(where):
            self._no("The `where=` keyword")
        if not self._kw_given(initial):
            return self._reduce(op, axis, keepdims)
        # NumPy's fold from a seed: ((v op x0) op x1) ... = v op (x0 dual x1 dual ...), dual = + for -, * for / (FieldArray._reduce_kw)
        cls = type(self)
        init = (initial if isinstance(initial, cls) else cls(initial)).reshape(())
        dual = L.OP_ADD if op in (L.OP_ADD, L.OP_SUB) else L.OP_MUL
        n_axis = self.size if axis is None else self.shape[axis % self.ndim]
        if n_axis == 0:  # an empty axis: the seed itself, in the result's shape
            lead = () if axis is None else tuple(d for i, d in enumerate(self.shape) if i != axis % self.ndim)
            if keepdims:
                lead = (1,) * self.ndim if axis is None else tuple(1 if i == axis % self.ndim else d for i, d in enumerate(self.shape))
            return cls._wrap(init._t.expand(lead + (cls._NL,)).contiguous())
        return self._binary(op, init, self._reduce(dual, axis, keepdims))

    # ---- `where=` on calls, np.convolve and @ as compositions of the element-wise kernels (slow and exact) ----
    def _ufunc_masked(self, ufunc, method, where, inputs, kwargs):
        cls = type(self)
        out = kwargs.pop("out", None)
        target = (out[0] if isinstance(out, tuple) else out) if out is not None else None
        if method == "outer":
            a, b = inputs
            inputs = (a.reshape(tuple(a.shape) + (1,) * b.ndim), b.reshape((1,) * a.ndim + tuple(b.shape)))
        shapes = [tuple(x.shape) if isinstance(x, FieldArray) else np.shape(x) for x in inputs]
        wshape = tuple(where.shape) if isinstance(where, torch.Tensor) else np.shape(where)
        full = tuple(np.broadcast_shapes(*shapes, wshape))
        mask = self._where_mask(where, full).unsqueeze(-1)
        one = cls.Ones(())._t
        safe = [cls._wrap(torch.where(mask, x._t.expand(full + (cls._NL,)), one).contiguous()) if isinstance(x, cls) else x for x in inputs]
        res = super().__array_ufunc__(ufunc, "__call__", *safe, **kwargs)
        if not isinstance(res, cls):
            self._no(f"The `where=` keyword of np.{ufunc.__name__}")
        tr = res._t.expand(full + (cls._NL,))
        if target is not None:
            if not isinstance(target, cls):
                raise TypeError(f"Argument 'out' must be a {cls.name} array (or a 1-tuple holding one), not {type(target)}.")
            if tuple(target.shape) != full:
                raise ValueError(f"non-broadcastable output operand with shape {tuple(target.shape)} doesn't match the broadcast shape {full}")
            target._t.copy_(torch.where(mask, tr, target._t))
            return target
        return cls._wrap(torch.where(mask, tr, torch.zeros((), dtype=torch.int64, device=tr.device)).contiguous())

    def _convolve(self, other):
        cls = type(self)
        na, nb = self.size, other.size
        acc = cls.Zeros(na + nb - 1)
        for i in range(na):  # out[i : i + nb] += a[i] * b
            acc._t[i:i + nb] = (cls._wrap(acc._t[i:i + nb].contiguous()) + self[i] * other)._t
        return acc

    def _matmul(self, other):
        # (..., M, K) @ (..., K, N): products of every (row, column) pair, then a left fold over K, as matmul_jit sums them
        a = self.reshape((1, self.size)) if self.ndim == 1 else self
        b = other.reshape((other.size, 1)) if other.ndim == 1 else other
        if a.shape[-1] != b.shape[-2]:
            raise ValueError(f"Operation 'matmul' requires the last dimension of 'A' to match the second-to-last dimension of 'B', not {tuple(self.shape)} and {tuple(other.shape)}.")
        prod = a.reshape(tuple(a.shape) + (1,)) * b.reshape(tuple(b.shape[:-2]) + (1,) + tuple(b.shape[-2:]))  # (..., M, K, N)
        r = prod._reduce(L.OP_ADD, -2, False)
        if self.ndim == 1 and other.ndim == 1:
            return r.reshape(tuple(r.shape[:-2]))
        if self.ndim == 1:
            return r.reshape(tuple(r.shape[:-2]) + (r.shape[-1],))
        if other.ndim == 1:
            return r.reshape(tuple(r.shape[:-1]))
        return r

    # ---- the complex128 bit-container trick of the two-limb fields does not extend to k limbs: the NumPy functions served here
    # work on the storage tensor with its limb axis kept last ----
    def _af_tens(self, v):
        self._no("This NumPy function")

    _af_seq = _af_wrap = _af_tens

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if self._kw_given(kwargs.get("where", None)) and method in ("__call__", "outer"):
            where = kwargs.pop("where")
            return self._ufunc_masked(ufunc, method, where, inputs, kwargs)
        if ufunc is np.matmul and method == "__call__":
            cls = type(self)
            if not (isinstance(inputs[0], cls) and isinstance(inputs[1], cls)):
                raise TypeError(f"Operation 'matmul' requires both operands to be arrays over {cls.name}.")
            return inputs[0]._matmul(inputs[1])
        if method == "outer":
            a, b = inputs
            cls = type(self)
            if not (isinstance(a, cls) and isinstance(b, cls)):
                self._no("ufunc.outer with a non-field operand")
            return getattr(np, ufunc.__name__)(a.reshape(tuple(a.shape) + (1,) * b.ndim), b.reshape((1,) * a.ndim + tuple(b.shape)))
        return super().__array_ufunc__(ufunc, method, *inputs, **kwargs)

    def __array_function__(self, func, types, args, kwargs):
        cls = type(self)
        x = args[0] if args else None
        nl = cls._NL

        def norm(axis, ndim):
            return int(axis) % ndim

        if func in (np.sum, np.prod) and isinstance(x, cls):
            op = L.OP_ADD if func is np.sum else L.OP_MUL
            axis = args[1] if len(args) > 1 else kwargs.get("axis", None)
            return x._reduce_kw(None, op, axis, bool(kwargs.get("keepdims", False)), kwargs.get("where", None), kwargs.get("initial", None))
        if func in (np.cumsum, np.cumprod) and isinstance(x, cls):
            op = L.OP_ADD if func is np.cumsum else L.OP_MUL
            axis = args[1] if len(args) > 1 else kwargs.get("axis", None)
            return x.reshape(-1)._accumulate(op, 0) if axis is None else x._accumulate(op, axis)
        if func is np.convolve and isinstance(x, cls) and isinstance(args[1], cls):
            if x.ndim != 1 or args[1].ndim != 1 or x.size == 0 or args[1].size == 0:
                raise ValueError("Operation 'convolve' requires non-empty 1-D arrays.")
            return x._convolve(args[1])
        if func is np.reshape:
            return x.reshape(args[1] if len(args) > 1 else kwargs.get("shape", kwargs.get("newshape")))
        if func is np.ravel:
            return x.reshape(-1)
        if func is np.copy:
            return x.copy()
        if func is np.transpose:
            axes = args[1] if len(args) > 1 else kwargs.get("axes", None)
            axes = list(reversed(range(x.ndim))) if axes is None else [norm(a, x.ndim) for a in axes]
            return cls._wrap(x._t.permute(*axes, x.ndim).contiguous())
        if func in (np.concatenate, np.stack):
            parts = [v if isinstance(v, cls) else cls(v) for v in args[0]]
            axis = args[1] if len(args) > 1 else kwargs.get("axis", 0)
            if func is np.concatenate:
                if axis is None:
                    return cls._wrap(torch.cat([v._t.reshape(-1, nl) for v in parts], dim=0))
                return cls._wrap(torch.cat([v._t for v in parts], dim=norm(axis, parts[0].ndim)))
            return cls._wrap(torch.stack([v._t for v in parts], dim=norm(axis, parts[0].ndim + 1)))
        if func is np.where and len(args) == 3:
            a, b = (v if isinstance(v, cls) else cls(v) for v in args[1:])
            cond = args[0].numpy() != 0 if isinstance(args[0], FieldArray) else np.asarray(args[0])
            ct = torch.as_tensor(cond, device=a._t.device)
            shape = tuple(torch.broadcast_shapes(tuple(ct.shape), a.shape, b.shape))
            return cls._wrap(torch.where(ct.broadcast_to(shape).unsqueeze(-1), a._t.expand(shape + (nl,)), b._t.expand(shape + (nl,))).contiguous())
        if func is np.array_equal:
            o = args[1]
            return bool(isinstance(o, cls) and tuple(o.shape) == tuple(x.shape) and np.all(x == o))
        if func is np.shape:
            return tuple(x.shape)
        if func is np.ndim:
            return x.ndim
        if func is np.size:
            return x.size
        self._no(f"np.{getattr(func, '__name__', func)}")

    def _unsupported(self, *a, **k):
        self._no("This operation")

    _reduceat = _at = _sqrt = log = is_square = _fft = _poly_evaluate = _dft = _unsupported
