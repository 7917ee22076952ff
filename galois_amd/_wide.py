"""
Field arrays over GF(q), 2^64 <= q < 2^128 -- the reference's dtype=object fields (src/galois/_fields/_ufunc.py:36-48,
_domains/_meta.py:39-41: arrays of Python integers through the pure-Python ufuncs).

Here the elements live on the GPU as two 64-bit limbs (an int64 tensor with a trailing axis of 2); the host sees Python
integers (`numpy()` returns dtype=object) exactly as with the reference.  The element-wise ufunc surface is covered --
add, subtract, multiply, divide, negative, reciprocal, power (arbitrary-size integer exponents), field * integer, square,
divmod / remainder, ==, indexing, reshaping and the data-movement NumPy functions -- through the kernels of csrc/gfa_wide.hip,
and since round 3 ufunc.reduce / accumulate (np.sum, prod, cumsum, cumprod), np.convolve, @ / np.matmul / dot / vdot / inner
(gfa_wide_reduce, gfa_wide_convolve, gfa_wide_matmul: what the reference's Sage fixtures for these fields pin), and since
round 5 reduceat / at, np.sqrt, np.log (Pohlig-Hellman + baby-step / giant-step), np.fft / ifft, polynomial evaluation, row
reduction / LU / PLU / inverse / determinant / solve (gfa_wide_row_reduce, gfa_wide_plu_decompose).  Codes (RS / BCH) over these
fields raise NotImplementedError.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib as L
from ._array import FieldArray, _device, _ptr, _stream

_M64 = (1 << 64) - 1


def wide_params(p: int, m: int, irr_int: int) -> tuple[int, list[int]]:
    """(kind, 27 parameter words) for gfa_wfield_create (layout: include/galois_amd.h)."""
    q = p**m
    w = [0] * 27
    if m == 1:
        kind = 1
        w[0], w[1] = p & _M64, p >> 64
        w[2] = (-pow(p, -1, 1 << 64)) & _M64
        r2 = pow(2, 256, p)
        w[3], w[4] = r2 & _M64, r2 >> 64
        w[5], w[6] = (p - 2) & _M64, (p - 2) >> 64
    elif p == 2:
        if m > 128:
            raise NotImplementedError(f"GF(2^{m}): binary fields are supported up to degree 128.")
        kind = 2
        w[0] = 2
        w[5], w[6] = (q - 2) & _M64, (q - 2) >> 64
        red = irr_int ^ (1 << m)
        w[7], w[8] = red & _M64, red >> 64
    else:
        if p >= 2**32 or m > 16:
            raise NotImplementedError(f"GF({p}^{m}): extension fields of order >= 2^64 need p < 2^32 and degree <= 16.")
        kind = 3
        w[0] = p
        itr = (q - 1) // (p - 1) - 1
        w[9], w[10] = itr & _M64, itr >> 64
        digits = []
        v = irr_int
        while v:
            digits.append(v % p)
            v //= p
        digits = digits[::-1]  # degree m .. 0
        for i, c in enumerate(digits[1:]):
            w[11 + i] = c
    return kind, w


def _split(values: np.ndarray, nl: int = 2) -> torch.Tensor:
    """Object array of Python integers in [0, 2^(64 nl)) -> int64 tensor (..., nl) of little-endian limbs."""
    flat = [int(v) for v in values.ravel()]
    limbs = np.empty((len(flat), nl), dtype=np.uint64)
    for i, v in enumerate(flat):
        for k in range(nl):
            limbs[i, k] = (v >> (64 * k)) & _M64
    return torch.from_numpy(limbs.view(np.int64).reshape(tuple(values.shape) + (nl,)))


class WideFieldArray(FieldArray):
    _wide_handle = None
    _limbed = True
    _NL = 2  # 64-bit limbs per element (galois_amd/_big.py: 4, 8 or 16 for orders above 2^128)
    _kernels = ("gfa_wide_binary", "gfa_wide_unary", "gfa_wide_power")

    def __init__(self, x, dtype=None, copy: bool = True):
        cls = type(self)
        if dtype is not None and np.dtype(dtype) != np.dtype(object):
            raise TypeError(f"{cls.name} arrays only support dtypes ['object'], not {np.dtype(dtype).name!r}.")
        self._np_dtype = np.dtype(object)
        if isinstance(x, FieldArray):
            if type(x) is not cls:
                raise TypeError(f"Cannot convert an array over {type(x).name} into an array over {cls.name}.")
            self._t = x._t.clone() if copy else x._t
            return
        if isinstance(x, torch.Tensor):
            raise TypeError(f"{cls.name} arrays are built from Python integers (two 64-bit limbs per element on the device).")
        arr = cls._verify_host(x)
        self._t = _split(np.asarray(arr, dtype=object), cls._NL).to(_device())

    @classmethod
    def _verify_host(cls, x) -> np.ndarray:
        """Element verification of array-likes (_fields/_array.py:129-180): Python integers in [0, order), kept as objects."""
        if isinstance(x, (int, np.integer)):
            arr = np.array(int(x), dtype=object)
        elif isinstance(x, (list, tuple, np.ndarray)):
            arr = np.array(x, dtype=object) if not isinstance(x, np.ndarray) else x
            if arr.dtype != object:
                if not np.issubdtype(arr.dtype, np.integer):
                    raise TypeError(f"{cls.name} arrays must have integer dtypes, not {arr.dtype}.")
                arr = arr.astype(object)
        else:
            raise TypeError(
                f"{cls.name} arrays can be created with scalars of type int, not {type(x)}."
                if np.isscalar(x) else f"{cls.name} arrays cannot be created from {type(x)}."
            )
        flat = arr.ravel()
        for v in flat:
            if not isinstance(v, (int, np.integer)):
                raise TypeError(f"{cls.name} arrays must have integer dtypes, not object elements of {type(v)}.")
        if flat.size and (min(int(v) for v in flat) < 0 or max(int(v) for v in flat) >= cls._order):
            raise ValueError(f"{cls.name} arrays must have elements in `0 <= x < {cls._order}`.")
        return arr

    @classmethod
    def _wrap(cls, t: torch.Tensor, np_dtype=None) -> "WideFieldArray":
        obj = object.__new__(cls)
        obj._t = t
        obj._np_dtype = np.dtype(object)
        return obj

    # ---- shape (the trailing limb axis is storage, not shape) ------------------------------------------------
    @property
    def shape(self):
        return tuple(self._t.shape[:-1])

    @property
    def ndim(self):
        return self._t.dim() - 1

    @property
    def size(self):
        return self._t.numel() // type(self)._NL

    def __len__(self):
        if self.ndim == 0:
            raise TypeError("len() of unsized object")
        return self._t.shape[0]

    @property
    def T(self):
        n = self.ndim
        return type(self)._wrap(self._t.permute(*reversed(range(n)), n).contiguous())

    def numpy(self) -> np.ndarray:
        nl = type(self)._NL
        host = self._t.cpu().numpy().view(np.uint64)
        out = np.empty(host.shape[:-1], dtype=object)
        flat = out.reshape(-1) if out.ndim else None
        limbs = host.reshape(-1, nl)
        join = lambda row: sum(int(row[k]) << (64 * k) for k in range(nl))
        if flat is None:
            return np.array(join(limbs[0]), dtype=object)
        for i in range(limbs.shape[0]):
            flat[i] = join(limbs[i])
        return out

    def __int__(self):
        if self.size != 1:
            raise TypeError("only size-1 arrays can be converted to Python scalars")
        return int(self.numpy().reshape(-1)[0])

    __index__ = __int__

    def copy(self):
        return type(self)._wrap(self._t.clone())

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        return type(self)._wrap(self._t.reshape(tuple(shape) + (type(self)._NL,)))

    def flatten(self):
        return type(self)._wrap(self._t.reshape(-1, type(self)._NL).clone())

    ravel = flatten

    def astype(self, dtype):
        if np.dtype(dtype) != np.dtype(object):
            raise TypeError(f"{type(self).name} arrays only support dtypes ['object'], not {np.dtype(dtype).name!r}.")
        return self.copy()

    def __getitem__(self, key):
        key = key if isinstance(key, tuple) else (key,)
        return type(self)._wrap(self._t[key + (Ellipsis, slice(None))] if Ellipsis not in key else self._t[key + (slice(None),)])

    def __setitem__(self, key, value):
        cls = type(self)
        v = value if isinstance(value, cls) else cls(value)
        key = key if isinstance(key, tuple) else (key,)
        self._t[key + ((Ellipsis, slice(None)) if Ellipsis not in key else (slice(None),))] = v._t

    def __eq__(self, other):
        cls = type(self)
        if not isinstance(other, cls):
            try:
                other = cls(other)
            except (TypeError, ValueError):
                return NotImplemented
        return (self._t == other._t).all(dim=-1).cpu().numpy()

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else ~r

    __hash__ = None

    # ---- constructors ----------------------------------------------------------------------------------------
    @classmethod
    def Zeros(cls, shape, dtype=None):
        shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(shape)
        return cls._wrap(torch.zeros(shape + (cls._NL,), dtype=torch.int64, device=_device()))

    @classmethod
    def Ones(cls, shape, dtype=None):
        z = cls.Zeros(shape)
        z._t[..., 0] = 1
        return z

    @classmethod
    def Identity(cls, size: int, dtype=None):
        z = cls.Zeros((size, size))
        z._t[torch.arange(size), torch.arange(size), 0] = 1
        return z

    @classmethod
    def Random(cls, shape=(), low: int = 0, high=None, seed=None, dtype=None):
        """Uniform in [low, high) from Python's `random` seeded like the reference's object-dtype path
        (_domains/_array.py:287-298 uses random.randint per element)."""
        import random

        high = cls._order if high is None else high
        if not 0 <= low < high <= cls._order:
            raise ValueError(f"Arguments must satisfy `0 <= low < high <= order`, not `0 <= {low} < {high} <= {cls._order}`.")
        shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(shape)
        rng = random.Random(None if seed is None else int(np.random.default_rng(seed).integers(0, 2**63)))
        n = int(np.prod(shape)) if shape else 1
        vals = np.array([rng.randint(low, high - 1) for _ in range(n)], dtype=object).reshape(shape)
        return cls(vals)

    @classmethod
    def Range(cls, start: int, stop: int, step: int = 1, dtype=None):
        return cls(np.array(list(range(start, stop, step)), dtype=object))

    # ---- arithmetic ------------------------------------------------------------------------------------------
    def _check_err(self, err):
        if int(err.item()) & L.DEVERR_ZERO_DIVISION:
            raise ZeroDivisionError("Cannot compute the multiplicative inverse of 0 in a Galois field.")

    @staticmethod
    def _bcast(a: torch.Tensor, b: torch.Tensor):
        nl = a.shape[-1]
        sa, sb = tuple(a.shape[:-1]), tuple(b.shape[:-1])
        shape = tuple(torch.broadcast_shapes(sa, sb))
        na, nb = a.numel() // nl, b.numel() // nl
        n = int(np.prod(shape)) if shape else 1
        ta, stra = (a.reshape(-1, nl).contiguous(), 0) if na == 1 and n != 1 else (a.expand(shape + (nl,)).contiguous(), 1)
        tb, strb = (b.reshape(-1, nl).contiguous(), 0) if nb == 1 and n != 1 else (b.expand(shape + (nl,)).contiguous(), 1)
        return ta, stra, tb, strb, shape, n

    def _binary(self, op, a, b):
        cls = type(self)
        ta, sa, tb, sb, shape, n = self._bcast(a._t, b._t)
        out = torch.empty(shape + (cls._NL,), dtype=torch.int64, device=ta.device)
        err = torch.zeros(1, dtype=torch.int32, device=ta.device) if op == L.OP_DIV else None
        L.check(getattr(L.lib(), cls._kernels[0])(cls._wide_handle, op, _ptr(ta), sa, _ptr(tb), sb, _ptr(out), n, _stream(),
                                                  _ptr(err) if err is not None else None), cls._kernels[0])
        if err is not None:
            self._check_err(err)
        return cls._wrap(out)

    def _unary(self, op):
        cls = type(self)
        t = self._t.contiguous()
        out = torch.empty_like(t)
        err = torch.zeros(1, dtype=torch.int32, device=t.device) if op == L.OP_RECIP else None
        L.check(getattr(L.lib(), cls._kernels[1])(cls._wide_handle, op, _ptr(t), _ptr(out), t.numel() // cls._NL, _stream(),
                                                  _ptr(err) if err is not None else None), cls._kernels[1])
        if err is not None:
            self._check_err(err)
        return cls._wrap(out)

    def _with_int(self, k, is_pow: bool):
        cls = type(self)
        if isinstance(k, torch.Tensor):
            k = k.cpu().numpy()
        if isinstance(k, (int, np.integer)):
            ks = np.array(int(k), dtype=object)
        elif isinstance(k, np.ndarray):
            if k.dtype != object and not np.issubdtype(k.dtype, np.integer):
                raise ValueError(f"Operation requires operands with type np.ndarray to have integer dtype, not {k.dtype}.")
            ks = k.astype(object)
        else:
            raise TypeError(f"{'The exponent' if is_pow else 'The integer multiplicand'} must be an integer or an integer np.ndarray, not {type(k)}.")
        if not is_pow:
            # field * integer = the integer reduced modulo the characteristic, as a field element (_ufunc.py:392-401)
            red = np.array([int(v) % cls._characteristic for v in ks.ravel()], dtype=object).reshape(ks.shape)
            return self._binary(L.OP_MUL, self, cls(red))
        qm1 = cls._order - 1
        flat = [int(v) for v in ks.ravel()]
        e_red = np.array([v % qm1 for v in flat], dtype=object).reshape(ks.shape)
        sign = torch.from_numpy(np.array([(v > 0) - (v < 0) for v in flat], dtype=np.int8).reshape(ks.shape)).to(self._t.device)
        te = _split(e_red, cls._NL).to(self._t.device)
        ta, sa, te2, se, shape, n = self._bcast(self._t, te)
        sg = sign.reshape(-1) if se == 0 else sign.expand(shape).contiguous()
        out = torch.empty(shape + (cls._NL,), dtype=torch.int64, device=ta.device)
        err = torch.zeros(1, dtype=torch.int32, device=ta.device)
        L.check(getattr(L.lib(), cls._kernels[2])(cls._wide_handle, _ptr(ta), sa, _ptr(te2), se, _ptr(sg), _ptr(out), n, _stream(), _ptr(err)),
                cls._kernels[2])
        self._check_err(err)
        return cls._wrap(out)

    # ---- NumPy functions: the data-movement branch of FieldArray.__array_function__ works on tensors with one entry per
    # element; here an element is the PAIR of limbs, re-viewed as one complex128 value (moved bit for bit, never computed on),
    # so that no axis argument can reach the limb axis ----
    def _af_tens(self, v) -> torch.Tensor:
        cls = type(self)
        w = v if isinstance(v, cls) else cls(v)
        return w._t.contiguous().view(torch.complex128).squeeze(-1)

    def _af_seq(self, ts):
        return list(ts), torch.complex128

    def _af_wrap(self, c: torch.Tensor):
        return type(self)._wrap(torch.view_as_real(c.contiguous()).view(torch.int64))

    # ---- ufunc.reduceat / ufunc.at: composed from the fold and the element-wise kernels (the reference runs them as object-dtype
    # loops over the same scalar functions, _domains/_ufunc.py:686-689) ----
    def _reduceat(self, op, indices, axis):
        idx = np.asarray(indices)
        if idx.ndim != 1 or not np.issubdtype(idx.dtype, np.integer):
            raise TypeError("Argument 'indices' of reduceat must be a 1-D integer array.")
        if self.ndim == 0:
            raise TypeError("cannot reduceat on a scalar")
        axis = axis % self.ndim
        n = self.shape[axis]
        if idx.size and (idx.min() < 0 or idx.max() >= n):
            raise IndexError(f"index {int(idx.max() if idx.max() >= n else idx.min())} out-of-bounds in reduceat [0, {n})")
        c = self._af_tens(self).movedim(axis, -1)
        ends = np.concatenate([idx[1:], [n]])
        pieces = []
        for s0, e0 in zip(idx.tolist(), ends.tolist()):
            if e0 > s0 + 1:
                pieces.append(self._af_tens(self._af_wrap(c[..., s0:e0].contiguous())._reduce(op, -1, False)))
            else:  # an empty or reversed slice yields a[indices[i]]
                pieces.append(c[..., s0])
        return self._af_wrap(torch.stack(pieces, dim=-1).movedim(-1, axis).contiguous())

    def _at(self, ufunc, indices, values):
        cls = type(self)
        idx = np.asarray(indices)
        if idx.dtype == bool or not np.issubdtype(idx.dtype, np.integer):
            raise TypeError("Argument 'indices' of ufunc.at must be an integer array (flat indices of a 1-D array or the first axis).")
        if self.ndim != 1:
            raise NotImplementedError("ufunc.at is implemented for 1-D field arrays.")
        n = self.shape[0]
        flat = idx.ravel().astype(np.int64)
        flat = np.where(flat < 0, flat + n, flat)
        if flat.size and (flat.min() < 0 or flat.max() >= n):
            raise IndexError(f"index out of bounds for axis 0 with size {n}")
        vals = None
        if values is not None:
            if ufunc is np.power or (ufunc is np.multiply and not isinstance(values, FieldArray)):
                vals = np.broadcast_to(np.asarray(values, dtype=object), idx.shape).ravel()
            else:
                v = values if isinstance(values, cls) else cls(values)
                vals = self._af_wrap(self._af_tens(v).broadcast_to(idx.shape).reshape(-1).contiguous())
        order = np.argsort(flat, kind="stable")
        sorted_idx = flat[order]
        group_start = np.r_[0, np.nonzero(np.diff(sorted_idx))[0] + 1] if flat.size else np.zeros(0, dtype=np.int64)
        occ_sorted = np.arange(flat.size) - np.repeat(group_start, np.diff(np.r_[group_start, flat.size]))
        occ = np.empty(flat.size, dtype=np.int64)
        occ[order] = occ_sorted
        for r in range(int(occ.max()) + 1 if flat.size else 0):  # round r: the r-th occurrence of every index (all distinct)
            sel = np.nonzero(occ == r)[0]
            ti = torch.from_numpy(flat[sel]).to(self._t.device)
            cur = cls._wrap(self._t[ti])
            if values is None:
                new = ufunc(cur)
            elif isinstance(vals, FieldArray):
                new = ufunc(cur, cls._wrap(vals._t[torch.from_numpy(sel).to(self._t.device)]))
            else:
                new = ufunc(cur, vals[sel])
            self._t[ti] = new._t
        return None

    # ---- squares and square roots (is_square: _fields/_array.py:1340-1410; sqrt: _domains/_calculate.py:758-832, the same
    # formulas as FieldArray._sqrt on the two-limb power kernel) ----
    def _eq_int(self, value: int) -> torch.Tensor:
        """Device bool mask: element == value (limb-wise)."""
        lim = torch.from_numpy(np.array([(value >> (64 * k)) & _M64 for k in range(type(self)._NL)], dtype=np.uint64).view(np.int64)).to(self._t.device)
        return (self._t == lim).all(dim=-1)

    def _select(self, mask: torch.Tensor, a: "WideFieldArray", b: "WideFieldArray") -> "WideFieldArray":
        return type(self)._wrap(torch.where(mask.unsqueeze(-1), a._t, b._t))

    def is_square(self):
        cls = type(self)
        if cls._characteristic == 2:
            r = np.ones(tuple(self.shape), dtype=bool)
        else:
            w = self._with_int((cls._order - 1) // 2, is_pow=True)
            r = (w._eq_int(1) | self._eq_int(0)).cpu().numpy()
        return bool(r) if r.ndim == 0 else r

    def _sqrt(self):
        """The formulas of FieldArray._sqrt (galois_amd/_array.py; reference _domains/_calculate.py:758-832) on two-limb elements."""
        cls = type(self)
        p, q = cls._characteristic, cls._order
        if p == 2:
            return self._with_int(2 ** (cls._degree - 1), is_pow=True)
        sq = self.is_square()
        if not np.all(sq):
            bad = self.numpy()[~np.asarray(sq)] if self.ndim else self.numpy()
            raise ArithmeticError(f"Input array has elements that are non-squares in {cls.name}.\n{bad}")
        zero = cls.Zeros(tuple(self.shape))
        if q % 4 == 3:
            roots = self._with_int((q + 1) // 4, is_pow=True)
        elif q % 8 == 5:
            d = self._with_int((q - 1) // 4, is_pow=True)
            r1 = self._with_int((q + 3) // 8, is_pow=True)
            four_a = self._with_int(4, is_pow=False)
            r2 = self._with_int(2, is_pow=False) * four_a._with_int((q - 5) // 8, is_pow=True)
            roots = self._select(d._eq_int(1), r1, self._select(d._eq_int(p - 1), r2, zero))
        else:
            # Tonelli-Shanks with a fixed non-square b (any non-square gives the same final min(root, -root))
            b = 2
            while cls(np.array(b, dtype=object)).is_square():
                b += 1
            n, s_ = q - 1, 0
            while n % 2 == 0:
                n >>= 1
                s_ += 1
            nz = ~self._eq_int(0)
            safe = self._select(nz, self, cls.Ones(tuple(self.shape)))
            a_inv = np.reciprocal(safe)
            c = cls(np.array(b, dtype=object))._with_int(n, is_pow=True)
            r = safe._with_int((n + 1) // 2, is_pow=True)
            for i in range(1, s_):
                dd = (r * r * a_inv)._with_int(2 ** (s_ - i - 1), is_pow=True)
                r = self._select(dd._eq_int(p - 1), r * c, r)
                c = c * c
            roots = self._select(nz, r, zero)
        neg = np.negative(roots)
        # np.minimum(roots, -roots) on the integer values: compare (hi, lo) as unsigned
        def ukey(t):
            return t ^ torch.iinfo(torch.int64).min
        neg_smaller = torch.zeros(tuple(self.shape), dtype=torch.bool, device=self._t.device)
        for k in range(cls._NL):  # limb 0 first: the comparison of a higher limb overrides it unless that limb is equal
            a, b = ukey(neg._t[..., k]), ukey(roots._t[..., k])
            neg_smaller = (a < b) | ((a == b) & neg_smaller)
        return self._select(neg_smaller, neg, roots)

    # ---- discrete logarithm: Pohlig-Hellman over the factorisation of q - 1 with baby-step / giant-step inside each prime
    # factor (the reference: log_pohlig_hellman, _domains/_calculate.py:700-755, on Python integers; any correct algorithm returns
    # the same unique i in [0, q - 1)).  Field arithmetic runs in the two-limb kernels on whole arrays; the host keeps the
    # bookkeeping integers (exponents, the CRT) and torch sorts / searches the baby-step keys. ----
    @classmethod
    def _log_tables(cls, base_int: int):
        cache = cls.__dict__.get("_log_cache")
        if cache is None:
            cache = {}
            cls._log_cache = cache
        if base_int in cache:
            return cache[base_int]
        from . import _numtheory as nt

        primes, mults = nt.factors(cls._order - 1)
        if max(primes) > 2**44:
            raise NotImplementedError(f"np.log over {cls.name}: q - 1 has the prime factor {max(primes)} (baby-step tables beyond 2^22 entries).")
        g = cls(np.array(base_int, dtype=object))
        tabs = []
        for r, e in zip(primes, mults):
            gamma = g._with_int((cls._order - 1) // r, is_pow=True)  # order r
            m = int(np.ceil(np.sqrt(r)))
            baby = gamma.reshape(()) ** np.array(list(range(m)), dtype=object)      # gamma^j
            lo = baby._t[:, 0]
            order = torch.argsort(lo, stable=True)
            giant = np.reciprocal(gamma._with_int(m, is_pow=True))                   # gamma^-m
            gpow = giant.reshape(()) ** np.array(list(range(m + 1)), dtype=object)  # gamma^(-m i)
            tabs.append((r, e, m, baby._t[order].contiguous(), order, gpow))
        cache[base_int] = (primes, mults, tabs)
        return cache[base_int]

    def log(self, base=None):
        cls = type(self)
        q1 = cls._order - 1
        if base is not None:
            b = base if isinstance(base, cls) else cls(base)
            if b.size != 1:
                raise NotImplementedError(f"np.log over {cls.name} takes one base for the whole array.")
            base_int = int(b)
            from . import _numtheory as nt

            primes, _ = nt.factors(q1)
            if base_int == 0 or any(int(b.reshape(()) ** (q1 // r)) == 1 for r in primes):
                raise ArithmeticError("The specified logarithm base is not a primitive element of the Galois field.")
        else:
            base_int = cls._primitive_element_int
        x = self.reshape(-1)
        n = x.size
        if n and bool(x._eq_int(0).any()):
            raise ArithmeticError("Cannot compute the discrete logarithm of 0 in a Galois field.")
        primes, mults, tabs = cls._log_tables(base_int)
        g = cls(np.array(base_int, dtype=object))
        residues = []  # per prime power: (modulus, list of n residues)
        for r, e, m, baby_sorted, order, gpow in tabs:
            xk = [0] * n
            cur = x  # x * g^(-xk) as digits are found
            for k in range(e):
                h = cur._with_int(q1 // r ** (k + 1), is_pow=True)  # in the subgroup of order r: h = gamma^(digit)
                digit = self._bsgs(h, m, baby_sorted, order, gpow, r)
                for i in range(n):
                    xk[i] += digit[i] * r**k
                if k + 1 < e:
                    corr = g.reshape(()) ** np.array([-(d * r**k) % q1 for d in digit], dtype=object)
                    cur = cur * corr
            residues.append((r**e, xk))
        out = np.empty(n, dtype=object)
        for i in range(n):
            acc, mod = 0, 1
            for modulus, xs in residues:  # CRT, one prime power at a time
                t = ((xs[i] - acc) * pow(mod, -1, modulus)) % modulus
                acc, mod = acc + mod * t, mod * modulus
            out[i] = acc % q1
        out = out.reshape(self.shape)
        return int(out) if out.ndim == 0 else out

    def _bsgs(self, h: "WideFieldArray", m: int, baby_sorted: torch.Tensor, order: torch.Tensor, gpow: "WideFieldArray", r: int):
        """digit[i] with gamma^digit[i] == h[i], 0 <= digit < r: h * gamma^(-m a) is looked up among the m baby steps gamma^j."""
        n = h.size
        digits = [None] * n
        chunk = max(1, (1 << 22) // (m + 1))
        keys = baby_sorted[:, 0].contiguous()
        for s0 in range(0, n, chunk):
            hs = type(self)._wrap(h._t[s0:s0 + chunk])
            prod = hs.reshape((hs.size, 1)) * gpow.reshape((1, m + 1))  # (rows, m + 1): h * gamma^(-m a)
            lo = prod._t[..., 0].contiguous()
            pos = torch.searchsorted(keys, lo).clamp(max=m - 1)
            hit = torch.zeros_like(lo, dtype=torch.bool)
            jidx = torch.zeros_like(lo)
            for probe in range(4):  # equal low limbs among the baby steps are astronomically rare; four neighbours are searched anyway
                pp = (pos + probe).clamp(max=m - 1)
                ok = (baby_sorted[pp] == prod._t).all(dim=-1) & ~hit
                jidx = torch.where(ok, order[pp], jidx)
                hit |= ok
            hit_c, j_c = hit.cpu().numpy(), jidx.cpu().numpy()
            for i in range(hs.size):
                a = np.flatnonzero(hit_c[i])
                if a.size == 0:
                    raise ArithmeticError("np.log: element outside the subgroup generated by the base (is the base primitive?)")
                digits[s0 + i] = (int(a[0]) * m + int(j_c[i, a[0]])) % r
        return digits

    # ---- np.fft.fft / ifft: the mixed-radix Cooley-Tukey recursion over the prime factors of n on whole-array kernels (matrix
    # product with the r x r DFT matrix, twiddle product), fft_jit.implementation (_domains/_function.py:246-384) computes the same
    # DFT with the same root of unity ----
    def _dft(self, omega: "WideFieldArray", n: int) -> "WideFieldArray":
        """X[k] = sum_j x[j] omega^(j k) along the LAST axis of a (..., n) array."""
        from . import _numtheory as nt
        from . import _linalg

        cls = type(self)
        if n == 1:
            return self
        primes, mults = nt.factors(n)
        r = int(primes[0])
        m = n // r
        lead = tuple(self.shape[:-1])
        # j = j1 * m + j2 (j1 < r), k = k1 + r * k2 (k1 < r): X[k] = sum_j2 [ (sum_j1 x[j1, j2] w_r^(j1 k1)) omega^(j2 k1) ] (omega^r)^(j2 k2)
        x2 = self.reshape(lead + (r, m))
        wr = omega ** m
        obj = lambda a: np.array([int(v) for v in np.ravel(a)], dtype=object).reshape(np.shape(a))
        F = wr.reshape(()) ** obj(np.outer(np.arange(r), np.arange(r)) % r)       # F[k1, j1] = w_r^(j1 k1)
        y = _linalg.matmul(F, x2)                                                  # (..., k1, j2)
        if m > 1:
            tw = omega.reshape(()) ** obj(np.outer(np.arange(r), np.arange(m)))    # omega^(k1 j2)
            y = y * tw
            y = y._dft(omega ** r, m)                                              # along j2 -> k2: (..., k1, k2)
        # output index k1 + r * k2: transpose the last two axes
        c = self._af_tens(y)
        return self._af_wrap(c.transpose(-1, -2).contiguous().reshape(lead + (n,)))

    def _fft(self, n, inverse: bool, scale: bool) -> "WideFieldArray":
        cls = type(self)
        if self.ndim != 1:
            raise ValueError("The FFT is only implemented on 1-D arrays.")
        length = self.size
        n = length if n is None else int(n)
        c = self._af_tens(self)
        if n < length:
            c = c[:n]
        elif n > length:
            c = torch.cat([c, self._af_tens(cls.Zeros(n - length))])
        x = self._af_wrap(c.contiguous())
        omega = cls(np.array(int(cls.primitive_root_of_unity(n)), dtype=object))  # ValueError if n does not divide q - 1
        if inverse:
            omega = np.reciprocal(omega)
        y = x._dft(omega, n)
        if scale:
            y = y / cls(np.array(n % cls._characteristic, dtype=object))
        return y

    def _poly_evaluate(self, x: "WideFieldArray") -> "WideFieldArray":
        """Horner evaluation of the polynomial whose coefficients (descending degree) are this 1-D array, at every element of x."""
        cls = type(self)
        co = self._t.contiguous()
        xt = x._t.contiguous()
        out = torch.empty_like(xt)
        L.check(L.lib().gfa_wide_poly_evaluate(cls._wide_handle, _ptr(co), self.size, _ptr(xt), _ptr(out), x.size, _stream()), "gfa_wide_poly_evaluate")
        return cls._wrap(out)

    # ---- ufunc.reduce / accumulate, np.convolve, @ : the reference runs them as object-dtype loops over the same scalar
    # kernels (_fields/_ufunc.py:36-48, _domains/_function.py:141-167, _domains/_linalg.py:286-308) ----
    def _fold(self, op, axis, accumulate: bool):
        cls = type(self)
        c = self._af_tens(self)  # one complex128 entry per element (two limbs), moved bit for bit
        if c.dim() == 0:
            raise TypeError("cannot reduce on a scalar")
        if axis is None:
            c2, lead = c.reshape(1, -1), ()
        else:
            axis = axis % c.dim()
            c2 = c.movedim(axis, -1)
            lead = tuple(c2.shape[:-1])
            c2 = c2.reshape(int(np.prod(lead, dtype=np.int64)), c2.shape[-1])  # (an explicit row count: the axis may be empty)
        c2 = c2.contiguous()
        n_outer, n_inner = c2.shape
        if n_inner == 0:
            # NumPy (and the reference's object-dtype loops): add / multiply have identities, subtract / divide do not
            if accumulate:
                return torch.empty((n_outer, 0), dtype=torch.complex128, device=c.device), lead, axis
            if op not in (L.OP_ADD, L.OP_MUL):
                raise ValueError("zero-size array to reduction operation which has no identity")
            limbs = torch.zeros((n_outer, 2), dtype=torch.int64, device=c.device)
            if op == L.OP_MUL:
                limbs[:, 0] = 1
            return torch.view_as_complex(limbs.view(torch.float64)), lead, axis
        out = torch.empty((n_outer, n_inner) if accumulate else (n_outer,), dtype=torch.complex128, device=c.device)
        err = torch.zeros(1, dtype=torch.int32, device=c.device) if op == L.OP_DIV else None
        L.check(L.lib().gfa_wide_reduce(cls._wide_handle, op, _ptr(c2), _ptr(out), n_outer, n_inner, 1 if accumulate else 0, _stream(),
                                        _ptr(err) if err is not None else None), "gfa_wide_reduce")
        if err is not None:
            self._check_err(err)
        return out, lead, axis

    def _reduce(self, op, axis, keepdims: bool):
        out, lead, axis_n = self._fold(op, axis, False)
        out = out.reshape(lead)
        if keepdims:
            out = out.reshape((1,) * self.ndim) if axis is None else out.unsqueeze(axis_n)
        return self._af_wrap(out)

    def _accumulate(self, op, axis):
        out, lead, axis_n = self._fold(op, axis, True)
        return self._af_wrap(out.reshape(lead + (out.shape[-1],)).movedim(-1, axis_n))

    def _convolve(self, other):
        cls = type(self)
        a, b = self._af_tens(self).contiguous(), self._af_tens(other).contiguous()
        out = torch.empty(a.numel() + b.numel() - 1, dtype=torch.complex128, device=a.device)
        L.check(L.lib().gfa_wide_convolve(cls._wide_handle, _ptr(a), a.numel(), _ptr(b), b.numel(), _ptr(out), _stream()), "gfa_wide_convolve")
        return self._af_wrap(out)

    @classmethod
    def _scalar(cls, op, a, b=0):
        x = cls(np.array(int(a), dtype=object))
        if op == L.OP_RECIP:
            return int(np.reciprocal(x))
        if op == L.OP_NEG:
            return int(-x)
        if op == L.OP_POW:
            return int(x ** int(b))
        y = cls(np.array(int(b), dtype=object))
        return int({L.OP_ADD: x + y, L.OP_SUB: x - y, L.OP_MUL: x * y, L.OP_DIV: x / y}[op])

    @classmethod
    def compile(cls, mode: str):
        if mode not in ("auto", "jit-calculate", "python-calculate"):
            raise ValueError(f"Argument 'mode' must be in ['auto', 'jit-calculate'] for {cls.name}, not {mode!r}.")
