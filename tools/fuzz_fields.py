"""Differential fuzzing of the element-wise path against the oracle: random fields (primes of every width, GF(2^m),
GF(p^m)), both modes, every legal dtype, random shapes with broadcasting and scalar operands; add, subtract, multiply,
divide, negative, reciprocal, power (negative exponents, zero bases), scalar multiplication, reductions.
Usage: python tools/fuzz_fields.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import galois_amd as ga
from oracle import gf_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 999
rng = np.random.default_rng(seed)
ORDERS = [2, 3, 5, 31, 251, 257, 65521, 65537, 7340033, 2**31 - 1, 4294967291, 2**61 - 1, 2**64 - 2**32 + 1, 2**2, 2**8, 2**11,
          2**16, 2**20, 2**32, 2**63, 3**2, 3**5, 7**3, 5**4, 251**3, 3**16, 31**6, 13**4, 3**10, 7**7, 997**2, 13**5]  # r05: packed-digit fields
t_end = time.time() + budget
count = 0


def rnd(q, shape):
    if q > 2**63:
        v = (rng.integers(0, 2**63, shape, dtype=np.uint64) * 2 + rng.integers(0, 2, shape, dtype=np.uint64)) % np.uint64(q)
    else:
        v = rng.integers(0, q, shape, dtype=np.uint64)
    v = np.asarray(v, dtype=np.uint64)
    if v.size > 3:
        flat = v.reshape(-1)
        flat[0] = 0; flat[1] = q - 1; flat[2] = 1
    return v


def eq(a, b, what):
    a = np.asarray(a)
    a = np.array([int(x) for x in a.ravel()], dtype=np.uint64).reshape(a.shape) if a.dtype == object else a.astype(np.uint64)
    assert np.array_equal(a, np.asarray(b).astype(np.uint64).reshape(a.shape)), what


while time.time() < t_end:
    q = int(ORDERS[rng.integers(0, len(ORDERS))])
    GF = ga.GF(q)
    F = O.OracleField(GF.characteristic, GF.degree, int(GF.irreducible_poly) if GF.degree > 1 else None, int(GF.primitive_element))
    modes = GF.ufunc_modes
    GF.compile(modes[int(rng.integers(0, len(modes)))])
    dts = [d for d in GF.dtypes if np.dtype(d) != np.dtype(object)]
    dt = dts[int(rng.integers(0, len(dts)))] if dts else None
    shape_a = tuple(int(s) for s in rng.integers(1, 40, int(rng.integers(1, 4))))
    bmode = rng.integers(0, 4)
    shape_b = shape_a if bmode == 0 else (() if bmode == 1 else (shape_a[-1:] if bmode == 2 else (shape_a[:-1] + (1,))))
    a, b = rnd(q, shape_a), rnd(q, shape_b)
    ga_, gb_ = (GF(a.astype(dt), dtype=dt), GF(b.astype(dt), dtype=dt)) if dt is not None else (GF([int(x) for x in a.ravel()]).reshape(shape_a), GF([int(x) for x in np.atleast_1d(b).ravel()]).reshape(shape_b) if shape_b else GF(int(b)))
    tag = (q, GF.ufunc_mode, np.dtype(dt).name if dt else "object", shape_a, shape_b)
    ab, bb = np.broadcast_arrays(a, b)
    eq((ga_ + gb_).numpy(), F.add(ab, bb), ("add",) + tag)
    eq((ga_ - gb_).numpy(), F.sub(ab, bb), ("sub",) + tag)
    eq((ga_ * gb_).numpy(), F.mul(ab, bb), ("mul",) + tag)
    eq((-ga_).numpy(), F.neg(a), ("neg",) + tag)
    bnz = np.where(b == 0, np.uint64(1), b)
    gbnz = GF(bnz.astype(dt), dtype=dt) if dt is not None else (GF([int(x) for x in np.atleast_1d(bnz).ravel()]).reshape(shape_b) if shape_b else GF(int(bnz)))
    eq((ga_ / gbnz).numpy(), F.div(ab, np.broadcast_to(bnz, ab.shape)), ("div",) + tag)
    eq(np.reciprocal(gbnz).numpy(), F.recip(bnz), ("recip",) + tag)
    if (b == 0).any():
        try:
            ga_ / gb_
            raise AssertionError(("expected ZeroDivisionError",) + tag)
        except ZeroDivisionError:
            pass
    e = rng.integers(-40, 200, shape_a)
    anz = np.where(a == 0, np.uint64(1), a)
    ganz = GF(anz.astype(dt), dtype=dt) if dt is not None else GF([int(x) for x in anz.ravel()]).reshape(shape_a)
    eq((ganz ** e).numpy(), F.pow(anz, e), ("pow",) + tag)
    eq((ga_ ** 3).numpy(), F.pow(a, np.full(shape_a, 3)), ("pow3",) + tag)
    kint = int(rng.integers(-1000, 1000))
    eq((ga_ * kint).numpy(), F.mul(a, np.full(shape_a, kint % GF.characteristic, dtype=np.uint64)), ("smul",) + tag)
    if len(shape_a) >= 2:
        red = np.add.reduce(ga_, axis=-1).numpy()
        want = a[..., 0]
        for j in range(1, shape_a[-1]):
            want = F.add(want, a[..., j])
        eq(red, want, ("add.reduce",) + tag)
        redm = np.multiply.reduce(ga_, axis=0).numpy()
        want = a[0]
        for j in range(1, shape_a[0]):
            want = F.mul(want, a[j])
        eq(redm, want, ("mul.reduce",) + tag)
    count += 1
print(f"fuzz_fields: {count} random (field, mode, dtype, shape) cases, every result identical to the oracle (seed {seed}, {budget:.0f} s)")
