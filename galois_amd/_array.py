"""
FieldArray: a finite-field array whose data lives in HBM (a PyTorch-ROCm tensor) and whose arithmetic runs in the
HIP kernels behind the C-ABI (include/galois_amd.h).

Host-side mirror of the reference's array domain -- same names, argument meaning and error behaviour for the hot
path (reference paths relative to /root/reference/src/galois):
  * construction / verification ........ _domains/_array.py:38-59, _fields/_array.py:129-215
  * Zeros/Ones/Range/Random/Identity .... _domains/_array.py:159-316
  * operator <-> ufunc routing .......... _domains/_ufunc.py:660-719 (UFuncMixin.__array_ufunc__) and the dispatcher
                                          rules of _ufunc.py:180-266, 384-508
  * np.fft.fft / np.fft.ifft ........... _domains/_function.py:177-212, 453-482
The reference's FieldArray *is* an np.ndarray (host memory); this one *wraps* device memory and implements the NumPy
protocols (__array_ufunc__, __array_function__, __array__), so `x * y`, `np.multiply(x, y)`, `np.reciprocal(x)`,
`x ** 3`, `np.add.reduce(x)`, `np.fft.fft(x)` keep working.  There is no CPU arithmetic path: without the HIP library
or a GPU every data operation raises.
"""
from __future__ import annotations

import ctypes
import threading
from typing import Any

import numpy as np
import torch

from . import _lib as L

# NumPy dtypes in the reference's preference order (_domains/_meta.py:19)
DTYPES = [np.uint8, np.uint16, np.uint32, np.int8, np.int16, np.int32, np.int64]

# storage: unsigned 16/32/64-bit patterns are kept in the signed torch dtype of the same width (all torch ops exist
# for those); the C-ABI only cares about the width
_TORCH_STORAGE = {1: torch.uint8, 2: torch.int16, 4: torch.int32, 8: torch.int64}
_NP_SIGNED = {1: np.uint8, 2: np.int16, 4: np.int32, 8: np.int64}
_GFA_DTYPE = {1: L.U8, 2: L.U16, 4: L.U32, 8: L.U64}


def _to_storage(t: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Field-element tensor -> another storage width.  uint16 / uint32 arrays live in same-width signed torch storage, so
    widening must undo the sign extension (elements are never negative)."""
    if t.dtype == dtype:
        return t
    wide = t.to(torch.int64)
    if t.dtype in (torch.int16, torch.int32):
        wide = wide & ((1 << (8 * t.element_size())) - 1)
    return wide.to(dtype)


def _unsigned_less(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a < b on unsigned bit patterns held in signed torch storage (uint8 storage compares natively)."""
    if a.dtype == torch.uint8:
        return a < b
    bias = torch.iinfo(a.dtype).min
    return (a ^ bias) < (b ^ bias)


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError(
            "galois_amd needs a ROCm GPU: field arrays live in HBM and all arithmetic runs in HIP kernels "
            "(there is deliberately no CPU fallback)."
        )
    return torch.device("cuda", torch.cuda.current_device())


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: torch.Tensor) -> int:
    return t.data_ptr()


# np functions overridden with field arithmetic (LinalgFunctionMixin._OVERRIDDEN_FUNCTIONS, _domains/_linalg.py:562-576)
_LINALG_FUNCTIONS = {np.dot: "dot", np.vdot: "vdot", np.inner: "inner", np.outer: "outer", np.linalg.det: "det",
                     np.linalg.matrix_rank: "matrix_rank", np.linalg.solve: "solve", np.linalg.inv: "inv"}


class FieldArrayMeta(type):
    """Class properties of a field (the reference keeps these on its metaclass too: _domains/_meta.py:112-203,
    _fields/_meta.py:73-752)."""

    def __repr__(cls) -> str:
        if getattr(cls, "_order", 0) == 0:
            return f"<class 'galois_amd.{cls.__name__}'>"
        return f"<class 'galois_amd.{cls.name}'>"

    @property
    def name(cls) -> str:
        if cls._degree == 1:
            return f"GF({cls._characteristic})"
        return f"GF({cls._characteristic}^{cls._degree})"

    @property
    def characteristic(cls) -> int:
        return cls._characteristic

    @property
    def degree(cls) -> int:
        return cls._degree

    @property
    def order(cls) -> int:
        return cls._order

    @property
    def irreducible_poly(cls):
        return cls._irreducible_poly

    @property
    def primitive_element(cls):
        return cls(cls._primitive_element_int) if torch.cuda.is_available() else cls._primitive_element_int

    @property
    def is_prime_field(cls) -> bool:
        return cls._degree == 1

    @property
    def is_extension_field(cls) -> bool:
        return cls._degree > 1

    @property
    def prime_subfield(cls):
        return cls._prime_subfield if cls._degree > 1 else cls

    @property
    def is_primitive_poly(cls) -> bool:
        return cls._is_primitive_poly

    @property
    def dtypes(cls) -> list:
        return list(cls._dtypes)

    @property
    def ufunc_mode(cls) -> str:
        if cls._handle is None:  # fields of order >= 2^64 (galois_amd/_wide.py): one device mode
            return cls._default_ufunc_mode
        return "jit-lookup" if L.lib().gfa_field_get_mode(cls._handle) == L.MODE_LOOKUP else "jit-calculate"

    @property
    def ufunc_modes(cls) -> list[str]:
        return list(cls._ufunc_modes)

    @property
    def default_ufunc_mode(cls) -> str:
        return cls._default_ufunc_mode

    @property
    def properties(cls) -> str:
        return (
            f"Galois Field:\n  name: {cls.name}\n  characteristic: {cls.characteristic}\n  degree: {cls.degree}\n"
            f"  order: {cls.order}\n  irreducible_poly: {cls._irreducible_poly}\n"
            f"  is_primitive_poly: {cls._is_primitive_poly}\n  primitive_element: {cls._primitive_element_str}"
        )


class FieldArray(metaclass=FieldArrayMeta):
    """An array over GF(p^m) resident on the GPU.  Instantiate through a class returned by `galois_amd.GF(...)`."""

    # filled in by the factory (galois_amd/_factory.py)
    _characteristic = 0
    _degree = 1
    _order = 0
    _irreducible_poly = None
    _primitive_element_int = 0
    _primitive_element_str = ""
    _is_primitive_poly = True
    _prime_subfield = None
    _dtypes: list = []
    _ufunc_modes: list = []
    _default_ufunc_mode = "jit-calculate"
    _handle = None
    _limbed = False        # True for fields of order >= 2^64: two 64-bit limbs per element in a trailing storage axis
    _object_dtype = False  # True when the reference would use dtype=object (order-1)^2 > int64 max: stored as uint64

    __array_priority__ = 100

    # ------------------------------------------------------------------------------------------------------------
    # construction
    # ------------------------------------------------------------------------------------------------------------
    def __init__(self, x: Any, dtype=None, copy: bool = True):
        cls = type(self)
        if cls._order == 0:
            raise NotImplementedError(
                "FieldArray is an abstract base class that cannot be directly instantiated. Instead, create a "
                "FieldArray subclass for GF(p^m) arithmetic using `GF = galois_amd.GF(p**m)` and instantiate an array "
                "using `x = GF(array_like)`."
            )
        np_dtype = cls._get_dtype(dtype)
        if isinstance(x, FieldArray):
            if type(x) is not cls:
                raise TypeError(f"Cannot convert an array over {type(x).name} into an array over {cls.name}.")
            t = x._t.clone() if copy else x._t
            self._np_dtype = x._np_dtype
            self._t = t
            if dtype is not None and np.dtype(np_dtype) != np.dtype(x._np_dtype):
                y = x.astype(np_dtype)
                self._t, self._np_dtype = y._t, y._np_dtype
            return
        if isinstance(x, torch.Tensor):
            self._t, self._np_dtype = cls._from_torch(x, np_dtype if dtype is not None else None, copy)
            return
        self._t, self._np_dtype = cls._from_host(x, np_dtype)

    @classmethod
    def _wrap(cls, t: torch.Tensor, np_dtype) -> "FieldArray":
        obj = object.__new__(cls)
        obj._t = t
        obj._np_dtype = np.dtype(np_dtype)
        return obj

    @classmethod
    def _get_dtype(cls, dtype):
        if dtype is None:
            return cls._dtypes[0]
        if dtype is object or (not cls._object_dtype and np.dtype(dtype) == np.dtype(object)):
            if cls._object_dtype:
                return np.object_
        if np.dtype(dtype) not in [np.dtype(d) for d in cls._dtypes]:
            raise TypeError(
                f"{cls.name} arrays only support dtypes {[np.dtype(d).name for d in cls._dtypes]}, not {np.dtype(dtype).name!r}."
            )
        return dtype

    @classmethod
    def _itemsize(cls, np_dtype) -> int:
        return 8 if np.dtype(np_dtype) == np.dtype(object) else np.dtype(np_dtype).itemsize

    @classmethod
    def _verify_host(cls, x) -> np.ndarray:
        """Element verification of array-likes (_fields/_array.py:129-180): integers in [0, order)."""
        if isinstance(x, (int, np.integer)):
            arr = np.array(int(x), dtype=object)
        elif isinstance(x, np.ndarray):
            arr = x
        elif isinstance(x, (list, tuple)):
            arr = np.array(x, dtype=object) if cls._object_dtype else np.array(x)
            if arr.dtype == object and not cls._object_dtype:
                arr = np.array(x, dtype=object)
        else:
            raise TypeError(
                f"{cls.name} arrays can be created with scalars of type int, not {type(x)}."
                if np.isscalar(x) else f"{cls.name} arrays cannot be created from {type(x)}."
            )
        if arr.dtype == object:
            flat = arr.ravel()
            for v in flat:
                if not isinstance(v, (int, np.integer)):
                    raise TypeError(f"{cls.name} arrays must have integer dtypes, not object elements of {type(v)}.")
            if flat.size and (min(int(v) for v in flat) < 0 or max(int(v) for v in flat) >= cls._order):
                raise ValueError(f"{cls.name} arrays must have elements in `0 <= x < {cls._order}`.")
            if cls._order <= 2**63:
                arr = arr.astype(np.int64)
            else:
                arr = np.array([int(v) for v in flat], dtype=np.uint64).reshape(arr.shape)
            return arr
        if not np.issubdtype(arr.dtype, np.integer):
            raise TypeError(f"{cls.name} arrays must have integer dtypes, not {arr.dtype}.")
        if arr.size:
            mn, mx = arr.min(), arr.max()
            if int(mn) < 0 or int(mx) >= cls._order:
                raise ValueError(f"{cls.name} arrays must have elements in `0 <= x < {cls._order}`, not {[int(mn), int(mx)]}.")
        return arr

    @classmethod
    def _from_host(cls, x, np_dtype):
        arr = cls._verify_host(x)
        size = cls._itemsize(np_dtype)
        shape = arr.shape  # np.ascontiguousarray promotes 0-D to 1-D: restore the shape afterwards
        if np.dtype(np_dtype) == np.dtype(object):
            store = np.ascontiguousarray(arr.astype(np.uint64)).view(np.int64)
        else:
            store = np.ascontiguousarray(arr.astype(np_dtype, copy=False)).view(_NP_SIGNED[size])
        t = torch.from_numpy(store.copy()).reshape(shape).to(_device())
        return t, np.dtype(np_dtype)

    @classmethod
    def _from_torch(cls, x: torch.Tensor, np_dtype, copy: bool):
        """Adopts a device tensor (zero-copy unless a cast is needed); verifies the range on the device."""
        if x.is_floating_point() or x.is_complex() or x.dtype == torch.bool:
            raise TypeError(f"{cls.name} arrays must have integer dtypes, not {x.dtype}.")
        if np_dtype is None:
            guess = {torch.uint8: np.uint8, torch.int8: np.int8, torch.int16: np.int16, torch.uint16: np.uint16,
                     torch.int32: np.int32, torch.uint32: np.uint32, torch.int64: np.int64, torch.uint64: np.uint64}[x.dtype]
            if cls._object_dtype:
                np_dtype = np.object_
            elif np.dtype(guess) in [np.dtype(d) for d in cls._dtypes]:
                np_dtype = guess
            else:
                raise TypeError(f"{cls.name} arrays only support dtypes {[np.dtype(d).name for d in cls._dtypes]}, not {x.dtype}.")
        size = cls._itemsize(np_dtype)
        # Range check FIRST, on the tensor as given and in its own signedness (narrowing would wrap out-of-range values
        # into the field, and comparing same-width signed storage with `order` wraps the scalar): work on the bit pattern
        # viewed as the same-width signed type.
        s_in = x.element_size()
        unsigned_in = x.dtype in (torch.uint8, torch.uint16, torch.uint32, torch.uint64)
        bits = x if x.dtype == torch.uint8 else x.view({1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[s_in])
        if x.numel():
            order, top, full = cls._order, 1 << (8 * s_in - 1), 1 << (8 * s_in)
            if x.dtype == torch.uint8:
                ok = True if order >= 256 else bool((bits < order).all().item())
            elif not unsigned_in:
                ok = bool((bits >= 0).all().item()) if order >= top else bool(((bits >= 0) & (bits < order)).all().item())
            elif order >= full:
                ok = True
            elif order <= top:
                ok = bool(((bits >= 0) & (bits < order)).all().item()) if order < top else bool((bits >= 0).all().item())
            else:  # top < order < full: patterns with the high bit set are valid up to order - 1
                ok = bool(((bits >= 0) | (bits < order - full)).all().item())
            if not ok:
                raise ValueError(f"{cls.name} arrays must have elements in `0 <= x < {cls._order}`.")
        if s_in == size:
            t = x.view(_TORCH_STORAGE[size]) if x.dtype != _TORCH_STORAGE[size] else x
        else:
            wide = bits.to(torch.int64)
            if unsigned_in and s_in < 8:
                wide = wide & ((1 << (8 * s_in)) - 1)  # undo the sign extension of the same-width signed view
            t = wide.to(_TORCH_STORAGE[size])
            copy = False
        if t.device.type != "cuda":
            t = t.to(_device())
            copy = False
        if copy:
            t = t.clone()
        if not t.is_contiguous():
            t = t.contiguous()
        return t, np.dtype(np_dtype)

    # ---- alternate constructors (_domains/_array.py:159-316) -----------------------------------------------
    @classmethod
    def Zeros(cls, shape, dtype=None) -> "FieldArray":
        np_dtype = cls._get_dtype(dtype)
        shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(shape)
        return cls._wrap(torch.zeros(shape, dtype=_TORCH_STORAGE[cls._itemsize(np_dtype)], device=_device()), np_dtype)

    @classmethod
    def Ones(cls, shape, dtype=None) -> "FieldArray":
        np_dtype = cls._get_dtype(dtype)
        shape = (shape,) if isinstance(shape, (int, np.integer)) else tuple(shape)
        return cls._wrap(torch.ones(shape, dtype=_TORCH_STORAGE[cls._itemsize(np_dtype)], device=_device()), np_dtype)

    @classmethod
    def Range(cls, start: int, stop: int, step: int = 1, dtype=None) -> "FieldArray":
        if not 0 <= start <= cls._order:
            raise ValueError(f"Argument 'start' must be within the field's order {cls._order}, not {start}.")
        if not 0 <= stop <= cls._order:
            raise ValueError(f"Argument 'stop' must be within the field's order {cls._order}, not {stop}.")
        np_dtype = cls._get_dtype(dtype)
        if cls._order > 2**63:
            return cls(np.array(list(range(start, stop, step)), dtype=object), dtype=dtype)
        return cls(np.arange(start, stop, step, dtype=np.int64), dtype=np_dtype)

    @classmethod
    def Random(cls, shape=(), low: int = 0, high: int | None = None, seed=None, dtype=None) -> "FieldArray":
        """Same bits as the reference for the same seed: rng.integers(low, high, shape, dtype) (_domains/_array.py:278-298)."""
        np_dtype = cls._get_dtype(dtype)
        high = cls._order if high is None else high
        if not 0 <= low < high <= cls._order:
            raise ValueError(f"Arguments must satisfy `0 <= low < high <= order`, not `0 <= {low} < {high} <= {cls._order}`.")
        rng = np.random.default_rng(seed)
        if np.dtype(np_dtype) != np.dtype(object):
            arr = rng.integers(low, high, shape, dtype=np_dtype)
        else:
            # object dtype in the reference: random.randint per element seeded from the generator (_array.py:283-296)
            import random as _random

            _seed = int(rng.integers(0, 2**63))
            _random.seed(_seed)
            n = int(np.prod(shape)) if shape != () else 1
            vals = [_random.randint(low, high - 1) for _ in range(n)]
            arr = np.array(vals, dtype=object).reshape(shape)
        return cls(arr, dtype=np_dtype)

    @classmethod
    def Identity(cls, size: int, dtype=None) -> "FieldArray":
        np_dtype = cls._get_dtype(dtype)
        return cls._wrap(torch.eye(size, dtype=_TORCH_STORAGE[cls._itemsize(np_dtype)], device=_device()), np_dtype)

    # ---- field-level helpers ------------------------------------------------------------------------------
    @classmethod
    def compile(cls, mode: str):
        """FieldArray.compile (_domains/_array.py:322-362).  "jit-lookup" = table kernels, "jit-calculate" = explicit
        arithmetic kernels, "auto" = the device default.  "python-calculate" does not exist here."""
        if mode not in ["auto"] + list(cls._ufunc_modes):
            raise ValueError(f"Argument 'mode' must be in {['auto'] + list(cls._ufunc_modes)} for {cls.name}, not {mode!r}.")
        m = {"auto": L.MODE_AUTO, "jit-lookup": L.MODE_LOOKUP, "jit-calculate": L.MODE_CALCULATE}[mode]
        L.check(L.lib().gfa_field_set_mode(cls._handle, m), "compile")

    @classmethod
    def _scalar(cls, op: int, a: int, b: int = 0) -> int:
        out = ctypes.c_uint64()
        rc = L.lib().gfa_scalar(cls._handle, op, int(a), int(b) & (2**64 - 1), ctypes.byref(out))
        if rc == L.ERR_INVALID and "division by zero" in L.last_error():
            raise ZeroDivisionError("Cannot compute the multiplicative inverse of 0 in a Galois field.")
        L.check(rc, "scalar arithmetic")
        return out.value

    @classmethod
    def primitive_root_of_unity(cls, n: int):
        """_fields/_array.py:1127-1187: alpha ** ((order - 1) // n)."""
        if not isinstance(n, (int, np.integer)):
            raise TypeError(f"Argument 'n' must be an instance of int, not {type(n)}.")
        if not 1 <= n < cls._order:
            raise ValueError(f"Argument 'n' must be in [1, {cls._order}), not {n}.")
        if not (cls._order - 1) % n == 0:
            raise ValueError(f"There are no primitive {n}-th roots of unity in {cls.name}.")
        return cls._root_of_unity_int(int(n))

    @classmethod
    def _root_of_unity_int(cls, n: int) -> int:
        e = (cls._order - 1) // n
        # exponent may exceed int64 only if order > 2^63: reduce by square-and-multiply on the host in that case
        if e < 2**63:
            return cls._scalar(L.OP_POW, cls._primitive_element_int, e)
        result, base = 1, cls._primitive_element_int
        while e:
            if e & 1:
                result = cls._scalar(L.OP_MUL, result, base)
            base = cls._scalar(L.OP_MUL, base, base)
            e >>= 1
        return result

    @classmethod
    def _tables(cls):
        """(EXP, LOG, ZECH_LOG, ZECH_E) as int64 arrays in the reference's layout (cls._EXP etc.)."""
        q = cls._order
        E = np.zeros(2 * q, dtype=np.int64)
        Lg = np.zeros(q, dtype=np.int64)
        Z = np.zeros(q, dtype=np.int64)
        ze = ctypes.c_int64()
        L.check(L.lib().gfa_field_tables(cls._handle, E.ctypes.data_as(L._i64p), Lg.ctypes.data_as(L._i64p),
                                         Z.ctypes.data_as(L._i64p), ctypes.byref(ze)), "tables")
        return E, Lg, Z, ze.value

    # ------------------------------------------------------------------------------------------------------------
    # ndarray-like surface
    # ------------------------------------------------------------------------------------------------------------
    @property
    def shape(self):
        return tuple(self._t.shape)

    @property
    def ndim(self) -> int:
        return self._t.dim()

    @property
    def size(self) -> int:
        return self._t.numel()

    @property
    def dtype(self):
        return self._np_dtype

    @property
    def device(self):
        return self._t.device

    @property
    def T(self):
        return type(self)._wrap(self._t.t().contiguous() if self._t.dim() == 2 else self._t.permute(*reversed(range(self._t.dim()))).contiguous(), self._np_dtype)

    def __len__(self) -> int:
        if self._t.dim() == 0:
            raise TypeError("len() of unsized object")
        return self._t.shape[0]

    def torch(self) -> torch.Tensor:
        """The underlying device tensor (storage dtype: same width as `dtype`; unsigned patterns in signed tensors)."""
        return self._t

    def numpy(self) -> np.ndarray:
        host = self._t.cpu().numpy()
        if self._np_dtype == np.dtype(object):
            u = host.view(np.uint64)
            out = np.empty(u.shape, dtype=object)
            flat = out.ravel()
            for i, v in enumerate(u.ravel()):
                flat[i] = int(v)
            return out
        return host.view(self._np_dtype)

    def __array__(self, dtype=None, copy=None):
        arr = self.numpy()
        return arr.astype(dtype) if dtype is not None else arr

    def __int__(self) -> int:
        if self.size != 1:
            raise TypeError("only size-1 arrays can be converted to Python scalars")
        return int(self.numpy().reshape(-1)[0])

    __index__ = __int__

    def __repr__(self) -> str:
        body = np.array2string(self.numpy(), separator=", ", threshold=50)
        order = f"{self._characteristic}^{self._degree}" if self._degree > 1 else f"{self._characteristic}"
        return f"GF({body}, order={order})"

    def __str__(self) -> str:
        return np.array2string(self.numpy(), threshold=50)

    def copy(self) -> "FieldArray":
        return type(self)._wrap(self._t.clone(), self._np_dtype)

    def reshape(self, *shape) -> "FieldArray":
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        return type(self)._wrap(self._t.reshape(tuple(shape)), self._np_dtype)

    def flatten(self) -> "FieldArray":
        return type(self)._wrap(self._t.reshape(-1).clone(), self._np_dtype)

    ravel = flatten

    def astype(self, dtype) -> "FieldArray":
        cls = type(self)
        np_dtype = cls._get_dtype(dtype)  # TypeError for dtypes the field does not allow (_domains/_array.py:445-451)
        size = cls._itemsize(np_dtype)
        if size == self._t.element_size():
            return cls._wrap(self._t.clone(), np_dtype)
        return cls._wrap(_to_storage(self._t, _TORCH_STORAGE[size]), np_dtype)

    def __getitem__(self, key) -> "FieldArray":
        key = self._convert_key(key)
        return type(self)._wrap(self._t[key], self._np_dtype)

    def __setitem__(self, key, value):
        cls = type(self)
        key = self._convert_key(key)
        if isinstance(value, FieldArray):
            if type(value) is not cls:
                raise TypeError(f"Cannot assign an array over {type(value).name} into an array over {cls.name}.")
            v = value._t.to(self._t.dtype) if value._t.dtype != self._t.dtype else value._t
        else:
            v = cls(value, dtype=self._np_dtype if self._np_dtype != np.dtype(object) else None)._t  # verifies the range
        self._t[key] = v

    @staticmethod
    def _convert_key(key):
        def conv(k):
            if isinstance(k, np.ndarray):
                return torch.from_numpy(k).to("cuda")
            if isinstance(k, FieldArray):
                raise IndexError("field arrays are not valid indices")
            return k
        if isinstance(key, tuple):
            return tuple(conv(k) for k in key)
        return conv(key)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __eq__(self, other):
        cls = type(self)
        if isinstance(other, FieldArray):
            if type(other) is not cls:
                return NotImplemented
            o = _to_storage(other._t, self._t.dtype)
            return (self._t == o).cpu().numpy()
        return self.numpy() == np.asarray(other)

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else ~r

    __hash__ = None

    # ------------------------------------------------------------------------------------------------------------
    # arithmetic
    # ------------------------------------------------------------------------------------------------------------
    def _gfa_dtype(self) -> int:
        return _GFA_DTYPE[self._t.element_size()]

    def _same_storage(self, other: "FieldArray") -> torch.Tensor:
        """Other operand's tensor in this array's storage width (the result keeps `self.dtype`, _ufunc.py:675)."""
        if other._t.element_size() == self._t.element_size():
            return other._t
        return _to_storage(other._t, self._t.dtype)

    @staticmethod
    def _broadcast(a: torch.Tensor, b: torch.Tensor):
        """Returns (a, stride_a, b, stride_b, out_shape) with strides in {0, 1} for the C-ABI."""
        if a.shape == b.shape:
            return a.contiguous(), 1, b.contiguous(), 1, a.shape
        out_shape = torch.broadcast_shapes(a.shape, b.shape)
        if b.numel() == 1:
            return a.expand(out_shape).contiguous() if a.shape != out_shape else a.contiguous(), 1, b.reshape(1), 0, out_shape
        if a.numel() == 1:
            return a.reshape(1), 0, b.expand(out_shape).contiguous() if b.shape != out_shape else b.contiguous(), 1, out_shape
        return a.expand(out_shape).contiguous(), 1, b.expand(out_shape).contiguous(), 1, out_shape

    def _check_err(self, err: torch.Tensor):
        if int(err.item()) & L.DEVERR_ZERO_DIVISION:
            raise ZeroDivisionError("Cannot compute the multiplicative inverse of 0 in a Galois field.")

    _out_tls = threading.local()  # .target: set by __array_ufunc__ while ONE element-wise kernel with `out=` runs (per thread)

    def _alloc_out(self, shape) -> torch.Tensor:
        """Result buffer of an element-wise kernel: the caller's `out=` tensor when it fits, else a fresh one."""
        tgt = getattr(FieldArray._out_tls, "target", None)
        if tgt is not None and tuple(tgt.shape) == tuple(shape) and tgt.dtype == self._t.dtype and tgt.device == self._t.device:
            FieldArray._out_tls.target = None
            return tgt
        return torch.empty(shape, dtype=self._t.dtype, device=self._t.device)

    @staticmethod
    def _extent(t: torch.Tensor):
        """[first byte, one past the last byte) a tensor view can touch."""
        if t.numel() == 0:
            return t.data_ptr(), t.data_ptr()
        span = sum((n - 1) * abs(st) for n, st in zip(t.shape, t.stride()))
        return t.data_ptr(), t.data_ptr() + (span + 1) * t.element_size()

    @classmethod
    def _may_write_in_place(cls, target: torch.Tensor, inputs) -> bool:
        """The kernels read each operand element exactly where they write the result element (pointers are __restrict__): an
        input may BE the target (same view) or be disjoint from it; a partial overlap (np.add(a[:-1], a[1:], out=a[1:])) must
        go through a fresh buffer, as NumPy's own overlap handling does."""
        lo, hi = cls._extent(target)
        for x in inputs:
            t = getattr(x, "_t", None)
            if t is None:
                continue
            a, b = cls._extent(t)
            if b <= lo or hi <= a:
                continue
            if a == lo and tuple(t.shape) == tuple(target.shape) and t.stride() == target.stride():
                continue
            return False
        return True

    def _binary(self, op: int, a: "FieldArray", b: "FieldArray") -> "FieldArray":
        cls = type(self)
        ta, tb = (a._t if a is self else self._same_storage(a)), (b._t if b is self else self._same_storage(b))
        ta, sa, tb, sb, shape = self._broadcast(ta, tb)
        out = self._alloc_out(shape)
        n = out.numel()
        err = torch.zeros(1, dtype=torch.int32, device=self._t.device) if op == L.OP_DIV else None
        L.check(L.lib().gfa_binary(cls._handle, op, _ptr(ta), sa, _ptr(tb), sb, _ptr(out), n, self._gfa_dtype(), _stream(),
                                   _ptr(err) if err is not None else None), "gfa_binary")
        if err is not None:
            self._check_err(err)
        return cls._wrap(out, self._np_dtype)

    def _unary(self, op: int) -> "FieldArray":
        cls = type(self)
        t = self._t.contiguous()
        out = self._alloc_out(t.shape)
        err = torch.zeros(1, dtype=torch.int32, device=t.device) if op == L.OP_RECIP else None
        L.check(L.lib().gfa_unary(cls._handle, op, _ptr(t), _ptr(out), t.numel(), self._gfa_dtype(), _stream(),
                                  _ptr(err) if err is not None else None), "gfa_unary")
        if err is not None:
            self._check_err(err)
        return cls._wrap(out, self._np_dtype)

    def _int_operand(self, k, what: str) -> torch.Tensor:
        """Integer scalar / integer ndarray -> device int64 tensor (np.power exponents, field*int multiplicands)."""
        if isinstance(k, (int, np.integer)):
            if not -(2**63) <= int(k) < 2**63:
                raise ValueError(f"{what} must fit in int64 on the device, not {k}.")
            return torch.tensor([int(k)], dtype=torch.int64, device=self._t.device).reshape(())
        if isinstance(k, np.ndarray):
            if k.dtype == object:
                k = np.array([int(v) for v in k.ravel()], dtype=np.int64).reshape(k.shape)
            if not np.issubdtype(k.dtype, np.integer):
                raise ValueError(f"Operation requires operands with type np.ndarray to have integer dtype, not {k.dtype}.")
            return torch.from_numpy(np.ascontiguousarray(k.astype(np.int64))).to(self._t.device)
        if isinstance(k, torch.Tensor) and not k.is_floating_point():
            return k.to(device=self._t.device, dtype=torch.int64)
        raise TypeError(f"{what} must be an integer or an integer np.ndarray, not {type(k)}.")

    def _with_int(self, k, is_pow: bool) -> "FieldArray":
        cls = type(self)
        tk = self._int_operand(k, "The exponent" if is_pow else "The integer multiplicand")
        ta, sa, tk, sk, shape = self._broadcast(self._t, tk)
        out = self._alloc_out(shape)
        if is_pow:
            err = torch.zeros(1, dtype=torch.int32, device=self._t.device)
            L.check(L.lib().gfa_power(cls._handle, _ptr(ta), sa, _ptr(tk), sk, _ptr(out), out.numel(), self._gfa_dtype(),
                                      _stream(), _ptr(err)), "gfa_power")
            self._check_err(err)
        else:
            L.check(L.lib().gfa_scalar_multiply(cls._handle, _ptr(ta), sa, _ptr(tk), sk, _ptr(out), out.numel(),
                                                self._gfa_dtype(), _stream()), "gfa_scalar_multiply")
        return cls._wrap(out, self._np_dtype)

    def _reduce(self, op: int, axis, keepdims: bool) -> "FieldArray":
        cls = type(self)
        t = self._t
        if t.dim() == 0:
            raise TypeError("cannot reduce on a scalar")
        if axis is None:
            t2 = t.reshape(1, -1)
            out_shape = ()
        else:
            axis = axis % t.dim()
            t2 = t.movedim(axis, -1).contiguous()
            out_shape = tuple(t2.shape[:-1])
            t2 = t2.reshape(-1, t2.shape[-1])
        t2 = t2.contiguous()
        out = torch.empty(t2.shape[0], dtype=t.dtype, device=t.device)
        err = torch.zeros(1, dtype=torch.int32, device=t.device) if op == L.OP_DIV else None
        L.check(L.lib().gfa_reduce(cls._handle, op, _ptr(t2), _ptr(out), t2.shape[0], t2.shape[1], self._gfa_dtype(),
                                   _stream(), _ptr(err) if err is not None else None), "gfa_reduce")
        if err is not None:
            self._check_err(err)
        out = out.reshape(out_shape)
        if keepdims:
            out = out.unsqueeze(axis if axis is not None else 0) if axis is not None else out.reshape((1,) * t.dim())
        return cls._wrap(out, self._np_dtype)

    def _reduceat(self, op: int, indices, axis: int) -> "FieldArray":
        """ufunc.reduceat: folds of a[indices[i] : indices[i+1]] along `axis` (the last one runs to the end; an empty or
        reversed slice yields a[indices[i]])."""
        cls = type(self)
        idx = np.asarray(indices)
        if idx.ndim != 1 or not np.issubdtype(idx.dtype, np.integer):
            raise TypeError("Argument 'indices' of reduceat must be a 1-D integer array.")
        t = self._t
        if t.dim() == 0:
            raise TypeError("cannot reduceat on a scalar")
        axis = axis % t.dim()
        n = t.shape[axis]
        if idx.size and (idx.min() < 0 or idx.max() >= n):
            raise IndexError(f"index {int(idx.max() if idx.max() >= n else idx.min())} out-of-bounds in reduceat [0, {n})")
        t2 = t.movedim(axis, -1).contiguous()
        lead = tuple(t2.shape[:-1])
        rows = int(np.prod(lead)) if lead else 1
        starts = np.asarray(idx, dtype=np.int64)
        ends = np.concatenate([starts[1:], [n]]).astype(np.int64)
        base = (np.arange(rows, dtype=np.int64) * n)[:, None]
        st = torch.from_numpy(np.ascontiguousarray((base + starts[None, :]).ravel())).to(t.device)
        en = torch.from_numpy(np.ascontiguousarray((base + ends[None, :]).ravel())).to(t.device)
        out = torch.empty(rows * idx.size, dtype=t.dtype, device=t.device)
        err = torch.zeros(1, dtype=torch.int32, device=t.device)
        L.check(L.lib().gfa_reduceat(cls._handle, op, _ptr(t2), _ptr(st), _ptr(en), st.numel(), _ptr(out), self._gfa_dtype(), _stream(),
                                     _ptr(err)), "gfa_reduceat")
        self._check_err(err)
        return cls._wrap(out.reshape(lead + (idx.size,)).movedim(-1, axis).contiguous(), self._np_dtype)

    def _at(self, ufunc, indices, values):
        """ufunc.at: unbuffered in-place a[indices] = op(a[indices], values); repeated indices are applied one after the other,
        as NumPy does: round r updates the r-th occurrence of every index (all distinct within a round)."""
        cls = type(self)
        idx = np.asarray(indices)
        if idx.dtype == bool or not np.issubdtype(idx.dtype, np.integer):
            raise TypeError("Argument 'indices' of ufunc.at must be an integer array (flat indices of a 1-D array or the first axis).")
        if self._t.dim() != 1:
            raise NotImplementedError("ufunc.at is implemented for 1-D field arrays.")
        n = self._t.shape[0]
        flat = idx.ravel().astype(np.int64)
        flat = np.where(flat < 0, flat + n, flat)
        if flat.size and (flat.min() < 0 or flat.max() >= n):
            raise IndexError(f"index out of bounds for axis 0 with size {n}")
        vals = None
        if values is not None:
            if ufunc is np.power or (ufunc is np.multiply and not isinstance(values, FieldArray)):
                vals = np.broadcast_to(np.asarray(values), idx.shape).ravel()
            else:
                v = values if isinstance(values, cls) else cls(values)
                vals = cls._wrap(torch.broadcast_to(self._same_storage(v), idx.shape).reshape(-1).contiguous(), self._np_dtype)
        # occurrence number of every entry within its index group (stable order)
        order = np.argsort(flat, kind="stable")
        sorted_idx = flat[order]
        group_start = np.r_[0, np.nonzero(np.diff(sorted_idx))[0] + 1] if flat.size else np.zeros(0, dtype=np.int64)
        occ_sorted = np.arange(flat.size) - np.repeat(group_start, np.diff(np.r_[group_start, flat.size]))
        occ = np.empty(flat.size, dtype=np.int64)
        occ[order] = occ_sorted
        for r in range(int(occ.max()) + 1 if flat.size else 0):
            sel = np.nonzero(occ == r)[0]
            ti = torch.from_numpy(flat[sel]).to(self._t.device)
            cur = cls._wrap(self._t[ti], self._np_dtype)
            if values is None:
                new = ufunc(cur)
            elif isinstance(vals, FieldArray):
                new = ufunc(cur, vals[torch.from_numpy(sel).to(self._t.device)])
            else:
                new = ufunc(cur, vals[sel])
            self._t[ti] = new._t
        return None

    def _accumulate(self, op: int, axis: int) -> "FieldArray":
        cls = type(self)
        t = self._t
        if t.dim() == 0:
            raise TypeError("cannot accumulate on a scalar")
        axis = axis % t.dim()
        t2 = t.movedim(axis, -1).contiguous()
        shape = t2.shape
        t2 = t2.reshape(-1, shape[-1])
        out = torch.empty_like(t2)
        err = torch.zeros(1, dtype=torch.int32, device=t.device) if op == L.OP_DIV else None
        L.check(L.lib().gfa_accumulate(cls._handle, op, _ptr(t2), _ptr(out), t2.shape[0], t2.shape[1], self._gfa_dtype(),
                                       _stream(), _ptr(err) if err is not None else None), "gfa_accumulate")
        if err is not None:
            self._check_err(err)
        return cls._wrap(out.reshape(shape).movedim(-1, axis).contiguous(), self._np_dtype)

    # ---- NumPy ufunc protocol (UFuncMixin.__array_ufunc__, _domains/_ufunc.py:660-713) ----------------------
    _UNARY_ONLY = (np.negative, np.reciprocal, np.square)
    _SINGLE_KERNEL_UFUNCS = (np.add, np.subtract, np.multiply, np.true_divide, np.floor_divide, np.negative, np.reciprocal,
                             np.power, np.square)

    # ---- the `where=` and `initial=` keywords (the reference forwards every keyword to the NumPy ufunc it built,
    # _domains/_ufunc.py:349, 364, 379, 403, 418; NumPy's semantics on the integer values are the contract) ----
    @staticmethod
    def _kw_given(v) -> bool:
        return not (v is None or v is True or v is np._NoValue)

    def _where_mask(self, where, shape) -> torch.Tensor:
        """`where` as a device bool tensor broadcast to `shape` (NumPy casts the mask with the 'safe' rule: bool only)."""
        if isinstance(where, torch.Tensor):
            m = where
            if m.dtype != torch.bool:
                raise TypeError(f"Cannot cast the 'where' mask from {m.dtype} to bool according to the rule 'safe'.")
        else:
            m = np.asarray(where)
            if m.dtype != np.bool_:
                raise TypeError(f"Cannot cast array data from {m.dtype!r} to dtype('bool') according to the rule 'safe'")
            m = torch.from_numpy(np.ascontiguousarray(m))
        return m.to(self._t.device).broadcast_to(tuple(shape))

    def _ufunc_masked(self, ufunc, method, where, inputs, kwargs):
        """ufunc(..., where=mask[, out=target]): positions where the mask is False are NOT computed -- they keep the target's
        value (zero in a fresh result; NumPy leaves them uninitialised) and can raise nothing.  Computed on the whole array with
        the field operands replaced by 1 at masked-out positions (1/1, 1**k, log 1, sqrt 1 are all defined), then blended."""
        cls = type(self)
        out = kwargs.pop("out", None)
        target = None
        if out is not None:
            target = out[0] if isinstance(out, tuple) else out
            if isinstance(out, tuple) and len(out) != 1:
                raise NotImplementedError("The `where=` keyword with several `out` arrays is not supported on device-resident arrays.")
        if method == "outer":
            a, b = inputs
            sa, sb = tuple(np.shape(a)), tuple(np.shape(b))
            rs = lambda v, shp: v.reshape(shp) if isinstance(v, FieldArray) else np.asarray(v).reshape(shp)
            inputs = (rs(a, sa + (1,) * len(sb)), rs(b, (1,) * len(sa) + sb))
        shapes = [tuple(x.shape) if isinstance(x, (FieldArray, torch.Tensor)) else np.shape(x) for x in inputs]
        wshape = tuple(where.shape) if isinstance(where, torch.Tensor) else np.shape(where)
        full = tuple(np.broadcast_shapes(*shapes, wshape))
        mask = self._where_mask(where, full)
        one = self._af_tens(cls.Ones(()))
        safe = []
        for x in inputs:
            if isinstance(x, cls):
                tx = self._af_tens(x).broadcast_to(full)
                safe.append(self._af_wrap(torch.where(mask, tx, _to_storage(one, tx.dtype))))
            else:
                safe.append(x)
        res = self.__array_ufunc__(ufunc, "__call__", *safe, **kwargs)
        parts = res if isinstance(res, tuple) else (res,)
        blended = []
        for r in parts:
            if isinstance(r, FieldArray):
                tr = self._af_tens(r).broadcast_to(full)
                if target is not None:
                    if not isinstance(target, cls):
                        raise TypeError(f"Argument 'out' must be a {cls.name} array (or a 1-tuple holding one), not {type(target)}.")
                    if tuple(target.shape) != full:
                        raise ValueError(f"non-broadcastable output operand with shape {tuple(target.shape)} doesn't match the broadcast shape {full}")
                    tt = self._af_tens(target)
                    new = self._af_wrap(torch.where(mask, _to_storage(tr, tt.dtype), tt))  # (a plain .to() would sign-extend uint16 / uint32 bit patterns)
                    target._t.copy_(new._t.reshape(target._t.shape))
                    blended.append(target)
                else:
                    blended.append(self._af_wrap(torch.where(mask, tr, torch.zeros((), dtype=tr.dtype, device=tr.device))))
            else:  # integer / boolean host results (np.log, comparisons)
                hr = np.broadcast_to(np.asarray(r), full)
                hm = mask.cpu().numpy()
                if target is not None:
                    if not isinstance(target, np.ndarray):
                        raise TypeError(f"Argument 'out' of {ufunc.__name__!r} must be a np.ndarray, not {type(target)}.")
                    np.copyto(target, hr, where=hm, casting="unsafe")
                    blended.append(target)
                else:
                    blended.append(np.where(hm, hr, np.zeros((), dtype=hr.dtype)))
        return tuple(blended) if isinstance(res, tuple) else blended[0]

    def _reduce_kw(self, ufunc, op: int, axis, keepdims: bool, where, initial) -> "FieldArray":
        """ufunc.reduce(x, axis, keepdims, where=mask, initial=v) with NumPy's meaning: masked-out elements do not take part,
        the fold starts from `initial` -- ((v op x0) op x1) ..., i.e. v op (x0 dual x1 dual ...) with dual = + for -, * for /."""
        cls = type(self)
        has_where, has_initial = self._kw_given(where), self._kw_given(initial)
        if not has_where and not has_initial:
            return self._reduce(op, axis, keepdims)
        if self.ndim == 0:
            raise TypeError("cannot reduce on a scalar")
        dual = L.OP_ADD if op in (L.OP_ADD, L.OP_SUB) else L.OP_MUL
        x = self
        if has_where:
            # the reference's ufuncs are numba.vectorize products WITHOUT an identity, except np.bitwise_xor (characteristic 2
            # add / subtract) and np.bitwise_and (GF(2) multiply): _fields/_ufunc.py:59-61, _fields/_gf2.py:93-96
            has_identity = (cls._characteristic == 2 and dual == L.OP_ADD) or (cls._order == 2 and op == L.OP_MUL)
            if not has_initial and not has_identity:
                raise ValueError(f"reduction operation '{ufunc.__name__}' does not have an identity, so to use a where mask one has to specify 'initial'")
            mask = self._where_mask(where, self.shape)
            tx = self._af_tens(self)
            fill = _to_storage(self._af_tens(cls.Zeros(()) if dual == L.OP_ADD else cls.Ones(())), tx.dtype)
            x = self._af_wrap(torch.where(mask, tx, fill))
        if not has_initial:
            return x._reduce(op, axis, keepdims)  # an identity exists: the filled-in elements are neutral
        init = initial if isinstance(initial, cls) else cls(initial)
        if init.ndim != 0 and init.size != 1:
            raise ValueError("Argument 'initial' of a reduction must be a scalar.")
        init = init.reshape(())
        n_axis = x.size if axis is None else x.shape[axis % x.ndim]
        if n_axis == 0:
            lead = () if axis is None else tuple(d for i, d in enumerate(x.shape) if i != axis % x.ndim)
            if keepdims:
                lead = (1,) * x.ndim if axis is None else tuple(1 if i == axis % x.ndim else d for i, d in enumerate(x.shape))
            return self._af_wrap(self._af_tens(init).broadcast_to(lead).clone())
        body = x._reduce(dual, axis, keepdims)
        return self._binary(op, init, body)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        cls = type(self)
        if self._kw_given(kwargs.get("where", None)) and method in ("__call__", "outer"):
            where = kwargs.pop("where")
            if self._kw_given(kwargs.get("initial", None)):
                raise TypeError(f"{ufunc.__name__}() got an unexpected keyword argument 'initial'")
            return self._ufunc_masked(ufunc, method, where, inputs, kwargs)
        out = kwargs.pop("out", None)
        if out is not None:
            # the kernel result is written into the caller's array (_ufunc.py:309-319); same field and shape required
            target = out[0] if isinstance(out, tuple) else out
            if not (isinstance(out, tuple) and len(out) == 1 or isinstance(out, FieldArray)) or not isinstance(target, cls):
                raise TypeError(f"Argument 'out' must be a {cls.name} array (or a 1-tuple holding one), not {type(target)}.")
            # ufuncs that are ONE element-wise kernel write straight into the caller's buffer (also when an input IS the
            # target: out[i] depends on element i only).  Composites (np.sqrt: several kernels that re-read their input),
            # other methods, partially overlapping operands, or a target of another width / layout are computed into a fresh
            # buffer and then stored.
            direct = (method == "__call__" and ufunc in cls._SINGLE_KERNEL_UFUNCS and target._t.is_contiguous()
                      and cls._may_write_in_place(target._t, inputs))
            FieldArray._out_tls.target = target._t if direct else None
            try:
                result = self.__array_ufunc__(ufunc, method, *inputs, **kwargs)
            finally:
                FieldArray._out_tls.target = None
            if not isinstance(result, cls) or tuple(result.shape) != tuple(target.shape):
                raise ValueError(f"Argument 'out' has shape {tuple(target.shape)} but the result has shape "
                                 f"{tuple(getattr(result, 'shape', ()))}.")
            if result._t.data_ptr() != target._t.data_ptr():
                target._t.copy_(_to_storage(result._t, target._t.dtype))
            return target
        # keywords: `casting` is overridden by the reference itself ("unsafe", _ufunc.py:678-680) and `dtype` only names the
        # intermediate type of a result that is cast back to the array's dtype (:686-687, :322-330) -- neither changes a
        # value here.  `where` masks: _ufunc_masked above (calls) and _reduce_kw (reductions, with `initial`); NumPy accepts
        # neither on accumulate / reduceat / at.
        if method != "reduce":
            for key in ("where", "initial"):
                if self._kw_given(kwargs.get(key, None)):
                    raise TypeError(f"{ufunc.__name__}.{method}() got an unexpected keyword argument {key!r}")
        unknown = set(kwargs) - {"where", "initial", "casting", "dtype", "order", "subok", "axis", "keepdims", "signature"}
        if unknown:
            raise TypeError(f"Unsupported keyword argument(s) {sorted(unknown)} for ufunc {ufunc.__name__!r} on {cls.name} arrays.")
        operands = list(range(len(inputs)))
        field_ops = [i for i in operands if isinstance(inputs[i], cls)]
        non_field = [i for i in operands if not isinstance(inputs[i], cls)]

        def same_field():
            if non_field:
                raise TypeError(
                    f"Operation {ufunc.__name__!r} requires both operands to be instances of {cls!r}, "
                    f"not {[type(inputs[i]) for i in operands]}."
                )

        if ufunc in (np.add, np.subtract, np.multiply, np.true_divide, np.floor_divide):
            op = {np.add: L.OP_ADD, np.subtract: L.OP_SUB, np.multiply: L.OP_MUL, np.true_divide: L.OP_DIV,
                  np.floor_divide: L.OP_DIV}[ufunc]
            if method == "__call__":
                if ufunc is np.multiply and non_field:
                    # field * integer = repeated addition with the integer reduced mod p (_ufunc.py:392-401)
                    k = inputs[non_field[0]]
                    if not isinstance(k, (int, np.integer, np.ndarray)):
                        raise TypeError(
                            f"Operation 'multiply' requires operands that are not {cls!r} arrays to be integers or an "
                            f"integer np.ndarray, not {type(k)}."
                        )
                    return inputs[field_ops[0]]._with_int(k, is_pow=False)
                same_field()
                return self._binary(op, inputs[0], inputs[1])
            if method == "reduce":
                same_field()
                return inputs[0]._reduce_kw(ufunc, op, kwargs.get("axis", 0), bool(kwargs.get("keepdims", False)),
                                            kwargs.get("where", None), kwargs.get("initial", None))
            if method == "accumulate":
                same_field()
                return inputs[0]._accumulate(op, kwargs.get("axis", 0))
            if method == "reduceat":
                if not isinstance(inputs[0], cls):
                    raise TypeError(f"Operation {ufunc.__name__!r} requires a {cls!r} array, not {type(inputs[0])}.")
                return inputs[0]._reduceat(op, inputs[1], kwargs.get("axis", 0))
            if method == "at":
                if not isinstance(inputs[0], cls):
                    raise TypeError(f"Operation {ufunc.__name__!r} requires a {cls!r} array, not {type(inputs[0])}.")
                return inputs[0]._at(ufunc, inputs[1], inputs[2] if len(inputs) > 2 else None)
            if method == "outer":
                same_field()
                a, b = inputs
                return self._binary(op, a.reshape(tuple(a.shape) + (1,) * b.ndim), b.reshape((1,) * a.ndim + tuple(b.shape)))
            raise NotImplementedError(f"Ufunc method {method!r} of {ufunc.__name__!r} is not implemented on the device.")
        if ufunc in (np.negative, np.reciprocal) and method == "at":
            return inputs[0]._at(ufunc, inputs[1], None)
        if ufunc is np.power and method == "at":
            return inputs[0]._at(ufunc, inputs[1], inputs[2])
        if ufunc in (np.negative, np.reciprocal):
            if method != "__call__":
                raise ValueError(
                    f"Ufunc method {method!r} is not supported on {ufunc.__name__!r}. Reduction methods are only "
                    "supported on binary functions."
                )
            return inputs[0]._unary(L.OP_NEG if ufunc is np.negative else L.OP_RECIP)
        if ufunc is np.positive and method == "__call__":
            return inputs[0].copy()
        if ufunc is np.power or ufunc is np.square:
            if ufunc is np.square:
                if method != "__call__":
                    raise ValueError(f"Ufunc method {method!r} is not supported on 'square'.")
                return inputs[0]._with_int(2, is_pow=True)
            if method in ("reduce", "accumulate", "reduceat"):
                raise ValueError(
                    f"Ufunc method {method!r} is not supported on 'power' because it takes inputs with type {cls!r} "
                    "array and integer array. Different types do not support reduction."
                )
            if not isinstance(inputs[0], cls):
                raise TypeError(f"Operation 'power' requires the first operand to be a {cls!r} array, not {type(inputs[0])}.")
            if isinstance(inputs[1], FieldArray):
                raise TypeError(f"Operation 'power' requires the second operand to be an integer array, not {type(inputs[1])}.")
            if method == "outer":
                # np.power.outer(x, ints): shape x.shape + ints.shape (used by Vandermonde, _fields/_array.py:371-373)
                base = inputs[0]
                tk = base._int_operand(np.asarray(inputs[1]) if not isinstance(inputs[1], torch.Tensor) else inputs[1], "The exponent")
                tb = base._t.reshape(tuple(base.shape) + (1,) * tk.dim())
                return cls._wrap(tb, base._np_dtype)._with_int(tk.reshape((1,) * base.ndim + tuple(tk.shape)), is_pow=True)
            if method != "__call__":
                raise NotImplementedError(f"Ufunc method {method!r} of 'power' is not implemented on the device.")
            return inputs[0]._with_int(inputs[1], is_pow=True)
        if ufunc is np.log:
            if method != "__call__":
                raise ValueError(f"Ufunc method {method!r} is not supported on 'log'.")
            return inputs[0].log()
        if ufunc is np.sqrt:
            if method != "__call__":
                raise ValueError(f"Ufunc method {method!r} is not supported on 'sqrt'.")
            return inputs[0]._sqrt()
        if ufunc is np.divmod and method == "__call__":
            same_field()
            q = self._binary(L.OP_DIV, inputs[0], inputs[1])
            return q, cls.Zeros(q.shape, dtype=self._np_dtype if self._np_dtype != np.dtype(object) else None)
        if ufunc is np.remainder and method == "__call__":
            same_field()
            shape = torch.broadcast_shapes(inputs[0].shape, inputs[1].shape)
            return cls.Zeros(tuple(shape), dtype=self._np_dtype if self._np_dtype != np.dtype(object) else None)
        if ufunc is np.matmul:
            if method != "__call__":
                raise ValueError(f"Ufunc method {method!r} is not supported on 'matmul'.")
            same_field()
            from . import _linalg
            return _linalg.matmul(inputs[0], inputs[1])
        if ufunc in (np.equal, np.not_equal) and method == "__call__":
            r = inputs[0].__eq__(inputs[1]) if isinstance(inputs[0], cls) else inputs[1].__eq__(inputs[0])
            return r if ufunc is np.equal else ~r
        raise NotImplementedError(
            f"The NumPy ufunc {ufunc.__name__!r} is not supported on {cls.name} arrays."
        )

    # ---- NumPy function protocol (FunctionMixin.__array_function__, _domains/_function.py:453-482) ---------
    # ---- array <-> "one entry per element" tensors for the data-movement branch of __array_function__ ----
    def _af_tens(self, v) -> torch.Tensor:
        cls = type(self)
        if isinstance(v, cls):
            return v._t
        return cls(v, dtype=self._np_dtype if self._np_dtype != np.dtype(object) else None)._t

    def _af_seq(self, ts):
        dt = max((t.dtype for t in ts), key=lambda d: torch.empty((), dtype=d).element_size())
        return [_to_storage(t, dt) for t in ts], dt

    def _af_wrap(self, t: torch.Tensor):
        cls = type(self)
        size = t.element_size()
        cands = [np.dtype(object)] if cls._object_dtype else [np.dtype(d) for d in cls._dtypes]
        np_dtype = self._np_dtype if cls._itemsize(self._np_dtype) == size else next(d for d in cands if cls._itemsize(d) == size)
        return cls._wrap(t.contiguous(), np_dtype)

    def _store_into(self, out, res, func):
        """`out=` of an array function: NumPy's contract -- the result is stored into the caller's array of the same field and shape
        (any storage width of the field), which is returned."""
        if out is None:
            return res
        cls = type(self)
        if not isinstance(out, cls) or not isinstance(res, cls):
            raise TypeError(f"Argument 'out' of np.{func.__name__} must be a {cls.name} array, not {type(out)}.")
        if tuple(out.shape) != tuple(res.shape):
            raise ValueError(f"Output array has shape {tuple(out.shape)}, the result of np.{func.__name__} has shape {tuple(res.shape)}.")
        out._t.copy_(_to_storage(res._t, out._t.dtype).reshape(out._t.shape))
        return out

    def __array_function__(self, func, types, args, kwargs):
        if func is np.fft.fft or func is np.fft.ifft:
            from ._ntt import _field_fft

            return _field_fft(args[0], inverse=func is np.fft.ifft, **{k: v for k, v in kwargs.items()},
                              **({"n": args[1]} if len(args) > 1 else {}))
        if func is np.convolve:
            from ._ntt import _field_convolve

            return _field_convolve(*args, **kwargs)
        if func in _LINALG_FUNCTIONS:
            from . import _linalg

            res = getattr(_linalg, _LINALG_FUNCTIONS[func])(*args)
            return self._store_into(kwargs.get("out"), res, func)  # `out=`: the reference copies the result into it (_domains/_linalg.py:269-274)
        # Reductions that NumPy implements through ufunc methods on the subclass (the reference reaches its field
        # kernels through ndarray.__array_function__ -> add.reduce / multiply.reduce ..., _domains/_function.py:476)
        x = args[0] if args else None
        cls = type(self)

        def kw(name, pos, default):
            return args[pos] if len(args) > pos else kwargs.get(name, default)

        def no_extra(*allowed):
            bad = [k for k, v in kwargs.items() if k not in allowed and v is not None and v is not np._NoValue]
            if bad:
                raise NotImplementedError(f"Keyword(s) {bad} of np.{func.__name__} are not supported on device-resident {cls.name} arrays.")

        if func in (np.sum, np.prod) and isinstance(x, cls):
            no_extra("axis", "keepdims", "a", "where", "initial")
            op = L.OP_ADD if func is np.sum else L.OP_MUL
            uf = np.add if func is np.sum else np.multiply
            axis = kw("axis", 1, None)
            keep = bool(kwargs.get("keepdims", False)) if kwargs.get("keepdims", False) is not np._NoValue else False
            where, initial = kwargs.get("where", None), kwargs.get("initial", None)
            if axis is None:
                if x._kw_given(where):
                    where = x._where_mask(where, x.shape).reshape(-1)
                r = x.reshape(-1)._reduce_kw(uf, op, 0, False, where, initial)
                return r.reshape((1,) * x.ndim) if keep else r
            return x._reduce_kw(uf, op, axis, keep, where, initial)
        if func in (np.cumsum, np.cumprod) and isinstance(x, cls):
            no_extra("axis", "a")
            axis = kw("axis", 1, None)
            op = L.OP_ADD if func is np.cumsum else L.OP_MUL
            return x.reshape(-1)._accumulate(op, 0) if axis is None else x._accumulate(op, axis)
        # Pure data movement: done on the device tensors, result re-viewed as the field (the reference's
        # _FUNCTIONS_REQUIRING_VIEW and the subclass-preserving ndarray functions).  _af_tens / _af_seq / _af_wrap map an
        # array to a tensor with ONE entry per field element and back (fields of order >= 2^64 override them: their storage
        # carries a limb axis that is not a data axis).
        def tens(v):
            if isinstance(v, FieldArray) and not isinstance(v, cls):
                raise TypeError(f"np.{func.__name__} cannot combine arrays over {type(v).name} and {cls.name}.")
            return self._af_tens(v)

        def seq(v):
            return self._af_seq([tens(e) for e in v])

        wrap = self._af_wrap

        if func is np.trace and isinstance(x, cls):
            no_extra("offset", "axis1", "axis2", "a")
            d = torch.diagonal(tens(x), offset=kw("offset", 1, 0), dim1=kw("axis1", 2, 0), dim2=kw("axis2", 3, 1))
            return wrap(d)._reduce(L.OP_ADD, -1, False)
        if func is np.diff and isinstance(x, cls):
            no_extra("n", "axis", "a")
            n, axis = kw("n", 1, 1), kw("axis", 2, -1)
            r = x
            for _ in range(int(n)):
                tr = tens(r)
                hi = wrap(tr.narrow(axis, 1, tr.shape[axis] - 1))
                lo = wrap(tr.narrow(axis, 0, tr.shape[axis] - 1))
                r = hi - lo
            return r

        if func in (np.concatenate, np.stack, np.vstack, np.hstack, np.dstack, np.column_stack):
            ts, _ = seq(args[0])
            if func is np.concatenate:
                axis = kw("axis", 1, 0)
                res = wrap(torch.cat([t.reshape(-1) for t in ts]) if axis is None else torch.cat(ts, dim=axis))
            elif func is np.stack:
                res = wrap(torch.stack(ts, dim=kw("axis", 1, 0)))
            else:
                f = {np.vstack: torch.vstack, np.hstack: torch.hstack, np.dstack: torch.dstack, np.column_stack: torch.column_stack}[func]
                res = wrap(f(ts))
            return self._store_into(kwargs.get("out"), res, func)
        if isinstance(x, cls):
            t = tens(x)
            if func is np.broadcast_to:
                return wrap(t.broadcast_to(tuple(np.atleast_1d(kw("shape", 1, None)).tolist())))
            if func is np.reshape:
                return x.reshape(kw("shape", 1, None) if "newshape" not in kwargs else kwargs["newshape"])
            if func is np.ravel:
                return x.reshape(-1)
            if func is np.transpose:
                axes = kw("axes", 1, None)
                return wrap(t.permute(*(reversed(range(t.dim())) if axes is None else axes)))
            if func is np.swapaxes:
                return wrap(t.transpose(args[1], args[2]))
            if func is np.moveaxis:
                return wrap(torch.movedim(t, args[1], args[2]))
            if func is np.squeeze:
                axis = kw("axis", 1, None)
                return wrap(t.squeeze() if axis is None else t.squeeze(axis))
            if func is np.expand_dims:
                axis = kw("axis", 1, None)
                for ax in sorted(np.atleast_1d(axis).tolist()):
                    t = t.unsqueeze(ax)
                return wrap(t)
            if func is np.flip:
                axis = kw("axis", 1, None)
                return wrap(torch.flip(t, dims=list(range(t.dim())) if axis is None else np.atleast_1d(axis).tolist()))
            if func is np.roll:
                shift, axis = kw("shift", 1, None), kw("axis", 2, None)
                return wrap(torch.roll(t.reshape(-1), int(shift)).reshape(t.shape) if axis is None
                            else torch.roll(t, shifts=tuple(np.atleast_1d(shift).tolist()), dims=tuple(np.atleast_1d(axis).tolist())))
            if func is np.tile:
                reps = tuple(np.atleast_1d(kw("reps", 1, None)).tolist())
                return wrap(t.reshape((1,) * (len(reps) - t.dim()) + tuple(t.shape)).repeat(*((1,) * (t.dim() - len(reps)) + reps)))
            if func is np.repeat:
                axis = kw("axis", 2, None)
                return wrap(torch.repeat_interleave(t.reshape(-1) if axis is None else t, int(kw("repeats", 1, None)), dim=0 if axis is None else axis))
            if func is np.copy:
                return x.copy()
            if func in (np.diag, np.diagonal):
                k = kw("k", 1, 0) if func is np.diag else kw("offset", 1, 0)
                if func is np.diag and t.dim() == 1:
                    return wrap(torch.diag(t, k))
                return wrap(torch.diagonal(t, offset=k, dim1=kw("axis1", 2, 0) if func is np.diagonal else 0,
                                           dim2=kw("axis2", 3, 1) if func is np.diagonal else 1))
            if func in (np.tril, np.triu):
                return wrap((torch.tril if func is np.tril else torch.triu)(t, kw("k", 1, 0)))
            if func is np.take:
                idx = torch.as_tensor(np.asarray(kw("indices", 1, None)), device=t.device)
                axis = kw("axis", 2, None)
                if axis is not None:
                    axis = int(axis) % t.dim()
                return wrap(t.reshape(-1)[idx] if axis is None else torch.index_select(t, axis, idx.reshape(-1)).reshape(
                    tuple(t.shape[:axis]) + tuple(idx.shape) + tuple(t.shape[axis + 1:])))
            if func is np.array_equal:
                o = args[1]
                return bool(isinstance(o, cls) and tuple(o.shape) == tuple(x.shape) and (x == o).all()) if isinstance(o, FieldArray) \
                    else bool(np.array_equal(x.numpy(), np.asarray(o)))
            if func in (np.count_nonzero, np.any, np.all, np.nonzero, np.argwhere, np.flatnonzero):
                return func(x.numpy() != 0, *args[1:], **kwargs)  # predicates on "element is zero": no field arithmetic involved
            if func is np.shape:
                return tuple(x.shape)
            if func is np.ndim:
                return x.ndim
            if func is np.size:
                return x.size if len(args) < 2 else x.shape[args[1]]
        # Ordering and editing by integer value, as NumPy does on the reference's ndarray subclass (np.sort / argsort / unique /
        # append / insert / delete are not in its _UNSUPPORTED_FUNCTIONS, _domains/_function.py:405-461)
        if func in (np.sort, np.argsort, np.unique) and isinstance(x, cls) and not cls._limbed:
            t = tens(x)
            bits = 8 * t.element_size()
            key = (t ^ torch.iinfo(torch.int64).min) if bits == 64 else (t.to(torch.int64) & ((1 << bits) - 1))  # unsigned order
            if func is np.unique:
                no_extra("ar", "return_inverse", "return_counts")
                inv, cnt = bool(kwargs.get("return_inverse", False)), bool(kwargs.get("return_counts", False))
                r = torch.unique(key.reshape(-1), sorted=True, return_inverse=inv, return_counts=cnt)
                u = r[0] if (inv or cnt) else r
                u = (u ^ torch.iinfo(torch.int64).min) if bits == 64 else u
                out = [wrap(u.to(t.dtype))]
                if inv:
                    out.append(r[1].reshape(tuple(t.shape)).cpu().numpy())
                if cnt:
                    out.append(r[-1].cpu().numpy())
                return out[0] if len(out) == 1 else tuple(out)
            no_extra("a", "axis", "kind", "stable")
            axis = kw("axis", 1, -1)
            if axis is None:
                key, t, axis = key.reshape(-1), t.reshape(-1), 0
            order = torch.argsort(key, dim=axis, stable=True)
            return order.cpu().numpy() if func is np.argsort else wrap(torch.gather(t, axis, order))
        if func in (np.append, np.insert, np.delete) and isinstance(x, cls):
            t = tens(x)
            axis = kwargs.get("axis", args[3] if func is np.insert and len(args) > 3 else (args[2] if func is not np.insert and len(args) > 2 else None))
            if func is np.append:
                ts, _ = seq([x, kw("values", 1, None)])
                return wrap(torch.cat([u.reshape(-1) for u in ts]) if axis is None else torch.cat(ts, dim=axis))
            if axis is None:
                t, axis = t.reshape(-1), 0
            axis = int(axis) % t.dim()
            n = t.shape[axis]
            if func is np.delete:
                keep = np.delete(np.arange(n), kw("obj", 1, None))
                return wrap(torch.index_select(t, axis, torch.as_tensor(keep, device=t.device)))
            # insert: where the new slots go -- and WHICH value lands in each -- is NumPy's own index arithmetic on a marker array
            # of distinct negative ids (it pairs values[i] with obj[i] also when obj is unsorted, and a scalar obj takes every
            # value, numpy/lib/_function_base_impl.py insert); the elements stay on the device
            obj = kw("obj", 1, None)
            (vals, t), _ = self._af_seq([tens(kw("values", 2, None)), t])
            tm = torch.movedim(t, axis, 0)
            while vals.dim() < t.dim():
                vals = vals.unsqueeze(0)  # ndmin = arr.ndim
            obj_arr = None if isinstance(obj, slice) else np.asarray(obj)
            if np.ndim(obj) == 0 and not isinstance(obj, slice):
                vm = vals  # NumPy moves the FIRST axis of the values to `axis`: it is the one that counts the new elements
            elif obj_arr is not None and obj_arr.dtype != bool and obj_arr.ndim == 1 and obj_arr.size == 1:
                vm = torch.movedim(vals, axis, 0)  # a size-1 index sequence takes NumPy's scalar path: EVERY value goes in at that index
            else:
                vm = torch.movedim(vals, axis, 0)
                count = len(np.insert(np.arange(n, dtype=np.int64), obj, 0)) - n
                if vm.shape[0] != count:
                    vm = vm.expand((count,) + tuple(vm.shape[1:]))
            ids = -(1 + np.arange(vm.shape[0], dtype=np.int64))
            marker = np.insert(np.arange(n, dtype=np.int64), obj, ids)
            slots = np.flatnonzero(marker < 0)
            vm = vm.expand((vm.shape[0],) + tuple(tm.shape[1:]))
            out = torch.empty((len(marker),) + tuple(tm.shape[1:]), dtype=tm.dtype, device=tm.device)
            out[torch.as_tensor(np.flatnonzero(marker >= 0), device=t.device)] = tm
            out[torch.as_tensor(slots, device=t.device)] = vm.to(tm.dtype)[torch.as_tensor(-marker[slots] - 1, device=t.device)]
            return wrap(torch.movedim(out, 0, axis))
        if func is np.where and len(args) == 3:
            ts, _ = seq(args[1:])
            cond = args[0].numpy() != 0 if isinstance(args[0], FieldArray) else np.asarray(args[0])
            return wrap(torch.where(torch.as_tensor(cond, device=ts[0].device), ts[0], ts[1]))
        # No silent host fallback: integer NumPy arithmetic on the elements would be WRONG in the field (the reference raises
        # for its _UNSUPPORTED_FUNCTIONS, _domains/_function.py:405-461; everything it supports beyond the list above goes
        # through ufuncs, which __array_ufunc__ serves)
        raise NotImplementedError(
            f"The NumPy function {func.__name__!r} is not supported on device-resident {cls.name} arrays. "
            "If you'd like to perform this operation on the data, call `array.numpy()` first and then call the function."
        )

    # ---- Python operators ---------------------------------------------------------------------------------
    def __add__(self, o): return np.add(self, o)
    def __radd__(self, o): return np.add(o, self)
    def __iadd__(self, o):
        r = np.add(self, o); self._t = r._t; return self
    def __sub__(self, o): return np.subtract(self, o)
    def __rsub__(self, o): return np.subtract(o, self)
    def __isub__(self, o):
        r = np.subtract(self, o); self._t = r._t; return self
    def __mul__(self, o): return np.multiply(self, o)
    def __rmul__(self, o): return np.multiply(o, self)
    def __imul__(self, o):
        r = np.multiply(self, o); self._t = r._t; return self
    def __truediv__(self, o): return np.true_divide(self, o)
    def __rtruediv__(self, o): return np.true_divide(o, self)
    def __itruediv__(self, o):
        r = np.true_divide(self, o); self._t = r._t; return self
    def __floordiv__(self, o): return np.floor_divide(self, o)
    def __rfloordiv__(self, o): return np.floor_divide(o, self)
    def __mod__(self, o): return np.remainder(self, o)
    def __divmod__(self, o): return np.divmod(self, o)
    def __neg__(self): return np.negative(self)
    def __pos__(self): return np.positive(self)
    def __pow__(self, o): return np.power(self, o)  # _ufunc.py:715-719

    def __matmul__(self, o): return np.matmul(self, o)
    def __rmatmul__(self, o): return np.matmul(o, self)

    # ---- orders, trace and norm (_fields/_array.py:1233-1360, 1759-1880) --------------------------------------------------
    def additive_order(self):
        """1 for the zero element, the characteristic otherwise."""
        cls = type(self)
        r = torch.where(self._t == 0, 1, cls._characteristic).cpu().numpy().astype(np.int64)
        return int(r) if r.ndim == 0 else r

    def multiplicative_order(self):
        """Smallest a > 0 with x**a == 1.  For every prime r | q - 1 the factor r is stripped from the candidate order while
        x**(order / r) == 1 -- element-wise masks on the device, the same value as the reference's divisor search / log formula."""
        cls = type(self)
        if bool((self._t == 0).any()):
            raise ArithmeticError("The multiplicative order of 0 is not defined.")
        from . import _numtheory as nt

        n = cls._order - 1
        order = np.full(tuple(self.shape), n, dtype=object)
        if n > 1:
            primes, mults = nt.factors(n)
            for r, e in zip(primes, mults):
                for _ in range(e):
                    # candidates still divisible by r: test x**(order / r) == 1 group by group of equal exponents
                    flat = order.ravel()
                    for val in sorted(set(int(v) for v in flat if int(v) % r == 0)):
                        ex = val // r
                        if ex >= 2**63:
                            continue
                        w = self._with_int(ex, is_pow=True)
                        hit = ((w._t == 1).cpu().numpy().ravel()) & np.array([int(v) == val for v in flat])
                        for i in np.nonzero(hit)[0]:
                            flat[i] = ex
                    order = flat.reshape(order.shape)
        if cls._order <= 2**63:
            order = order.astype(np.int64)
        return int(order) if order.ndim == 0 else order

    def field_trace(self) -> "FieldArray":
        """Tr(x) = sum of x**(p**i), i < m, an element of the prime subfield."""
        cls = type(self)
        sub = cls.prime_subfield
        if cls._degree == 1:
            return self.copy()
        acc = self
        conj = self
        for _ in range(1, cls._degree):
            conj = conj._with_int(cls._characteristic, is_pow=True)
            acc = acc + conj
        return sub._wrap(_to_storage(acc._t, _TORCH_STORAGE[sub._itemsize(sub._get_dtype(None))]), sub._get_dtype(None))

    def field_norm(self) -> "FieldArray":
        """N(x) = x**((q - 1) / (p - 1)), an element of the prime subfield."""
        cls = type(self)
        sub = cls.prime_subfield
        if cls._degree == 1:
            return self.copy()
        w = self._with_int((cls._order - 1) // (cls._characteristic - 1), is_pow=True)
        return sub._wrap(_to_storage(w._t, _TORCH_STORAGE[sub._itemsize(sub._get_dtype(None))]), sub._get_dtype(None))

    # ---- vector-space view over the prime subfield (_fields/_array.py:383-491) ----------------------------------------
    def vector(self, dtype=None) -> "FieldArray":
        """FieldArray.vector: shape (...,) over GF(p^m) -> shape (..., m) over GF(p), degree m-1 first."""
        cls = type(self)
        sub = cls.prime_subfield
        np_dtype = sub._get_dtype(dtype)
        m = cls._degree
        t = self._t.contiguous()
        out = torch.empty(tuple(t.shape) + (m,), dtype=_TORCH_STORAGE[sub._itemsize(np_dtype)], device=t.device)
        L.check(L.lib().gfa_vector(cls._handle, 1, _ptr(t), self._gfa_dtype(), _ptr(out), _GFA_DTYPE[out.element_size()], t.numel(),
                                   _stream()), "gfa_vector")
        return sub._wrap(out, np_dtype)

    @classmethod
    def Vector(cls, array, dtype=None) -> "FieldArray":
        """FieldArray.Vector: length-m vectors over GF(p) (last axis, degree m-1 first) -> elements of GF(p^m)."""
        np_dtype = cls._get_dtype(dtype)
        sub = cls.prime_subfield
        x = array if isinstance(array, FieldArray) and type(array) is sub else sub(array)
        if x.ndim == 0 or not x.shape[-1] == cls._degree:
            raise ValueError(
                f"Argument 'array' must have last dimension equal to the field extension dimension {cls._degree}, "
                f"not {x.shape[-1] if x.ndim else ()}."
            )
        t = x._t.contiguous()
        out = torch.empty(tuple(t.shape[:-1]), dtype=_TORCH_STORAGE[cls._itemsize(np_dtype)], device=t.device)
        L.check(L.lib().gfa_vector(cls._handle, 0, _ptr(t), _GFA_DTYPE[t.element_size()], _ptr(out), _GFA_DTYPE[out.element_size()],
                                   out.numel(), _stream()), "gfa_vector")
        return cls._wrap(out, np_dtype)

    # ---- discrete logarithm, squares and square roots --------------------------------------------------------------
    def log(self, base=None):
        """FieldArray.log (_fields/_array.py:2127-2200): integer array i with base**i == self; base defaults to the
        field's primitive element.  Returns a host int64 array (a Python int for 0-D input), like the reference."""
        cls = type(self)
        if base is None:
            t, sa, tb, sb, out_shape = self._t.contiguous(), 1, None, 0, tuple(self._t.shape)
        else:
            b = base if isinstance(base, cls) else cls(base)
            t, sa, tb, sb, out_shape = self._broadcast(self._t, self._same_storage(b))
        if cls._order > 2**20 and not getattr(cls, "_log_prepared", False):
            # no LOG table: the device runs Pohlig-Hellman and needs the factorisation of q - 1 (host number theory, once)
            from . import _numtheory as nt

            primes, mults = nt.factors(cls._order - 1)
            pa = (ctypes.c_uint64 * len(primes))(*primes)
            ma = (ctypes.c_uint32 * len(primes))(*mults)
            L.check(L.lib().gfa_log_prepare(cls._handle, pa, ma, len(primes)), "gfa_log_prepare")
            cls._log_prepared = True
        out = torch.empty(out_shape, dtype=torch.int64, device=t.device)
        err = torch.zeros(1, dtype=torch.int32, device=t.device)
        L.check(L.lib().gfa_log(cls._handle, _ptr(t), sa, _ptr(tb) if tb is not None else None, sb, _ptr(out), out.numel(),
                                self._gfa_dtype(), _stream(), _ptr(err)), "gfa_log")
        e = int(err.item())
        if e & L.DEVERR_LOG_ZERO:
            raise ArithmeticError("Cannot compute the discrete logarithm of 0 in a Galois field.")
        if e & L.DEVERR_LOG_BASE:
            raise ArithmeticError("The specified logarithm base is not a primitive element of the Galois field.")
        res = out.cpu().numpy()
        if cls._order > 2**63:  # logarithms up to q - 2 do not fit int64: Python integers, like the reference's object arrays
            u = res.view(np.uint64)
            obj = np.array([int(v) for v in u.ravel()], dtype=object).reshape(u.shape)
            return int(obj) if obj.ndim == 0 else obj
        return int(res) if res.ndim == 0 else res

    def is_square(self):
        """FieldArray.is_square (_fields/_array.py:1340-1410): x is a square iff x == 0 or x^((q-1)/2) == 1; every
        element of a characteristic-2 field is a square.  Returns a host bool array."""
        cls = type(self)
        if cls._characteristic == 2:
            r = np.ones(tuple(self.shape), dtype=bool)
        else:
            w = self._with_int((cls._order - 1) // 2, is_pow=True)
            r = ((w._t == 1) | (self._t == 0)).cpu().numpy()
        return bool(r) if r.ndim == 0 else r

    def _sqrt(self) -> "FieldArray":
        """np.sqrt (sqrt_binary / sqrt, _domains/_calculate.py:758-832): the smaller of the two roots, as an integer."""
        cls = type(self)
        p, q = cls._characteristic, cls._order
        if p == 2:
            return self._with_int(2 ** (cls._degree - 1), is_pow=True)
        sq = self.is_square()
        if not np.all(sq):
            bad = self.numpy()[~np.asarray(sq)] if self.ndim else self.numpy()
            raise ArithmeticError(f"Input array has elements that are non-squares in {cls.name}.\n{bad}")
        if q % 4 == 3:
            roots = self._with_int((q + 1) // 4, is_pow=True)
        elif q % 8 == 5:
            d = self._with_int((q - 1) // 4, is_pow=True)
            r1 = self._with_int((q + 3) // 8, is_pow=True)
            four_a = self._with_int(4, is_pow=False)
            r2 = self._with_int(2, is_pow=False) * four_a._with_int((q - 5) // 8, is_pow=True)
            t = torch.where(d._t == 1, r1._t, torch.where(d._t == p - 1, r2._t, torch.zeros_like(r1._t)))
            roots = cls._wrap(t, self._np_dtype)
        else:
            # Tonelli-Shanks with a fixed non-square b (any non-square gives the same final min(root, -root))
            b = 2
            while cls(b).is_square():
                b += 1
            n, s_ = q - 1, 0
            while n % 2 == 0:
                n >>= 1
                s_ += 1
            tt = n
            minus_one = p - 1  # -1 is the constant p - 1 of the prime subfield (`d == p - 1`, _calculate.py:826)
            nz = self._t != 0
            safe = cls._wrap(torch.where(nz, self._t, torch.ones_like(self._t)), self._np_dtype)
            a_inv = np.reciprocal(safe)
            c = cls._scalar(L.OP_POW, b, tt)
            r = safe._with_int((tt + 1) // 2, is_pow=True)
            for i in range(1, s_):
                dd = (r * r * a_inv)._with_int(2 ** (s_ - i - 1), is_pow=True)
                rc = r * cls(c)
                r = cls._wrap(torch.where(dd._t == minus_one, rc._t, r._t), self._np_dtype)
                c = cls._scalar(L.OP_MUL, c, c)
            roots = cls._wrap(torch.where(nz, r._t, torch.zeros_like(r._t)), self._np_dtype)
        neg = np.negative(roots)
        # np.minimum(roots, -roots) on the integer representations (field elements are non-negative)
        small = torch.where(_unsigned_less(neg._t, roots._t), neg._t, roots._t)
        return cls._wrap(small, self._np_dtype)

    @classmethod
    def Vandermonde(cls, element, rows: int, cols: int, dtype=None) -> "FieldArray":
        """FieldArray.Vandermonde (_fields/_array.py:334-374): V[i, j] = (element**i)**j."""
        if not isinstance(element, (int, np.integer, cls)):
            raise TypeError(f"Argument 'element' must be an instance of (int, np.integer, {cls.name}), not {type(element)}.")
        for name, v in (("rows", rows), ("cols", cols)):
            if not isinstance(v, (int, np.integer)):
                raise TypeError(f"Argument {name!r} must be an instance of int, not {type(v)}.")
        if not rows > 0:
            raise ValueError(f"Argument 'rows' must be non-negative, not {rows}.")
        if not cols > 0:
            raise ValueError(f"Argument 'cols' must be non-negative, not {cols}.")
        element = element if isinstance(element, cls) else cls(int(element), dtype=dtype)
        if not element.ndim == 0:
            raise ValueError(f"Argument 'element' must be element scalar, not {element.ndim}-D.")
        v = element ** np.arange(0, rows)
        return np.power.outer(v, np.arange(0, cols))

    # ---- linear algebra methods (_fields/_array.py:1412-1760) -------------------------------------------------------
    def row_reduce(self, ncols=None, eye: str = "left"):
        from . import _linalg
        return _linalg.row_reduce(self, ncols=ncols, eye=eye)

    def lu_decompose(self):
        from . import _linalg
        return _linalg.lu_decompose(self)

    def plu_decompose(self):
        from . import _linalg
        return _linalg.plu_decompose(self)

    def row_space(self):
        from . import _linalg
        return _linalg.row_space(self)

    def column_space(self):
        from . import _linalg
        return _linalg.column_space(self)

    def left_null_space(self):
        from . import _linalg
        return _linalg.left_null_space(self)

    def null_space(self):
        from . import _linalg
        return _linalg.null_space(self)
