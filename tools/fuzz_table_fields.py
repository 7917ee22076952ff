"""Differential fuzzing of the LDS-table element-wise kernels (csrc/gfa_elementwise_mid.hip) against the oracle's lookup
ufuncs: random fields of 257 .. 65536 elements on uint16 storage, array lengths around the thresholds that switch kernels
(2^17, 2^19) and odd lengths up to 2^21, aligned and misaligned views, scalar operands on either side, zeros sprinkled in,
every operation incl. one exponent for the whole array and one per element, in `jit-lookup` and `auto`.
Usage: python tools/fuzz_table_fields.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import galois_amd as ga
from oracle import gf_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 999
rng = np.random.default_rng(seed)
ORDERS = [257, 2**9, 2**10, 2**13, 2**14, 2**15, 2**16, 3**6, 3**7, 3**8, 3**9, 3**10, 5**4, 5**5, 5**6, 7**4, 7**5, 11**3, 11**4, 13**4,
          17**3, 23**3, 37**3, 251**2, 181**2, 509, 8191, 8209, 16381, 32749, 32771, 65521]
LENGTHS = [2**17 - 9, 2**17, 2**17 + 5, 2**19 - 1, 2**19, 2**19 + 13, 300_007, 1_000_003, 2**21 + 3, 4_200_001]
t_end = time.time() + budget
count = 0
fields = {}


def eq(got, want, what):
    g = got.numpy().astype(np.uint64)
    if not np.array_equal(g, want):
        bad = np.nonzero(g != want)[0]
        raise AssertionError((what, "mismatches", bad.size, bad[:8].tolist()))


while time.time() < t_end:
    q = int(ORDERS[rng.integers(0, len(ORDERS))])
    if q not in fields:
        GF = ga.GF(q)
        fields[q] = (GF, O.OracleField(GF.characteristic, GF.degree, int(GF.irreducible_poly) if GF.degree > 1 else None,
                                       int(GF.primitive_element), lookup=True))
    GF, F = fields[q]
    mode = ("jit-lookup", "auto")[int(rng.integers(0, 2))]
    GF.compile(mode)
    n = int(LENGTHS[rng.integers(0, len(LENGTHS))])
    off = int(rng.integers(0, 2)) * int(rng.integers(1, 9))          # 0: aligned; else a view starting 1..8 elements in
    a = rng.integers(0, q, n + off, dtype=np.uint64)
    b = rng.integers(0, q, n + off, dtype=np.uint64)
    for arr in (a, b):
        arr[rng.integers(0, n + off, 50)] = 0
        arr[rng.integers(0, n + off, 20)] = q - 1
        arr[rng.integers(0, n + off, 20)] = 1
    bnz = np.where(b == 0, np.uint64(1), b)
    A, B, Bnz = (GF(v.astype(np.uint16), dtype=np.uint16)[off:] for v in (a, b, bnz))
    a, b, bnz = a[off:], b[off:], bnz[off:]
    tag = (q, mode, n, off)
    eq(A + B, F.add(a, b), ("add",) + tag)
    eq(A - B, F.sub(a, b), ("sub",) + tag)
    eq(-A, F.neg(a), ("neg",) + tag)
    eq(A * B, F.mul(a, b), ("mul",) + tag)
    eq(A / Bnz, F.div(a, bnz), ("div",) + tag)
    eq(np.reciprocal(Bnz), F.recip(bnz), ("recip",) + tag)
    k = int(rng.integers(0, n))
    full = lambda v: np.full(n, v, dtype=np.uint64)
    eq(A * B[k], F.mul(a, full(b[k])), ("mul by scalar",) + tag)
    eq(A[k] - B, F.sub(full(a[k]), b), ("scalar - array",) + tag)
    eq(A[k] / Bnz, F.div(full(a[k]), bnz), ("scalar / array",) + tag)
    eq(A / Bnz[k], F.div(a, full(bnz[k])), ("array / scalar",) + tag)
    e = int(rng.integers(-2**62, 2**62)) if rng.integers(0, 2) else int(rng.integers(-300, 300))
    eq(Bnz ** e, F.pow(bnz, np.full(n, e, dtype=np.int64)), ("pow", e) + tag)
    ev = rng.integers(-2**40, 2**40, n)
    ev[: n // 2] = rng.integers(-50, 1000, n // 2)
    eq(Bnz ** ev, F.pow(bnz, ev), ("pow each",) + tag)
    for fn, what in ((lambda: A / B, "div"), (lambda: np.reciprocal(B), "recip"), (lambda: A ** -3, "pow")):
        if (b == 0).any() if what != "pow" else (a == 0).any():
            try:
                fn()
                raise AssertionError(("expected ZeroDivisionError", what) + tag)
            except ZeroDivisionError:
                pass
    GF.compile("auto")
    count += 1
print(f"fuzz_table_fields: {count} random (field, mode, length, alignment) cases, every result identical to the oracle (seed {seed}, {budget:.0f} s)")
