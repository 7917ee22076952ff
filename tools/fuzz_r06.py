"""Differential fuzzing of the round-5 / round-6 paths against the oracle, sized so that the ORACLE is not the bottleneck
(tools/fuzz_r05.py drew 2^22-point transforms and fresh fields every case and managed 25 cases a minute).

    python tools/fuzz_r06.py [seconds] [seed]

Fields are created once (a pool of NTT primes of every bit length 17 .. 32, the table / binary / extension fields of the masked
ufunc cases); a case then costs a few kernel launches and one small oracle call:
  * ntt      random power-of-two length 2^2 .. 2^12 (one large draw, up to 2^20, about once a minute), random primitive root (an odd
             power of the field's), random batch, rows at the magnitude limit; forward against the oracle on three rows, batched
             against single, scaled inverse as a round trip.  The pool covers every kernel family: GF(65537) (shift twiddles),
             p < 2^26 / 2^28 / 2^29 (signed Montgomery, the three BMAX classes), [2^29, 2^32) (lazy Shoup);
  * ntt_grouped  GF(65537), 2^10 .. 2^15 points in batches of at least 2^22 points (G = 2^16 / n transforms per workgroup of the
             one-pass kernel), batch sizes that are not multiples of G;
  * ntt16    the one-workgroup 2^16-point kernels (r06: GF(65537) with the first twiddles formed in registers and the early
             requests; generic p < 2^29 with the early requests): batches of 64 .. 300 transforms, forward and scaled inverse;
  * ext32    extension fields of 2^15 .. 2^20 elements on uint32 arrays (two-word packed sums of GF(3^11) / GF(3^12), digit-table products,
             GF(p^2) / GF(p^3) quotients and reciprocals by the norm / Cramer's rule): every operation with tails, misaligned views, scalars;
  * long     one long row: np.add.reduce / np.multiply.reduce against a halving tree of the oracle's op, .accumulate against its recurrence (streaming folds,
             segmented scans), np.convolve over extension fields through Karatsuba planes checked by evaluation at random points;
  * convolve random lengths 1500 .. 6000 (the CRT route) over pool primes, five coefficients against Python integers;
  * where    masked ufunc calls / reductions on random fields, uint16 / uint32 results blended into WIDER `out` arrays;
  * wide     (1 case in 25) the two-limb identities of fuzz_r05.py.
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
from oracle import gf_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(seed)
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
counts = {"ntt": 0, "ntt_large": 0, "ntt16": 0, "ntt_grouped": 0, "ext32": 0, "long": 0, "convolve": 0, "where": 0, "wide": 0}
LONG = {}  # fields of the long-row folds / scans and of the plane convolutions (r06)
LONG_ORDERS = [2**8, 3**5, 31, 65537, 7340033, 2**16, 7**3, 2**20, 2**4, 5**4, 3**10]


def find_prime(bits, adic):
    """the largest prime c * 2^adic + 1 with exactly `bits` bits (a smaller 2-adicity when there is none)"""
    while adic > 1:
        c = ((1 << bits) - 1) >> adic
        while c > 0:
            p = c * (1 << adic) + 1
            if p <= (1 << (bits - 1)):
                break
            if p < (1 << bits) and ga.is_prime(p):
                return p
            c -= 1
        adic -= 1
    raise RuntimeError(bits)


# one NTT prime per bit length (2-adicity >= 16 where the length allows), plus the named ones
POOL = []
for p in [65537, 7340033, 469762049, 2013265921, 2130706433, 3221225473, 4293918721] + [find_prime(b, min(b - 3, 20)) for b in range(17, 33)]:
    GF = ga.GF(p)
    adic = ((p - 1) & -(p - 1)).bit_length() - 1
    POOL.append((p, GF, O.OracleField(p, 1, None, int(GF.primitive_element)), adic))
P16 = [e for e in POOL if e[0] < 2**29 and e[3] >= 16]
EXT32 = {}  # extension fields on uint32 arrays: two-word packed sums, digit-table products, quotients by the norm / Cramer's rule (r06)
EXT32_ORDERS = [257**2, 509**2, 997**2, 251**2, 191**2, 37**3, 41**3, 67**3, 97**3, 101**3, 3**11, 3**12, 7**7, 5**8, 13**5, 31**4, 2**17, 2**19, 2**20,
                251**3, 1021**3, 1447**3, 1031**2, 8191**2, 37813**2]  # the last six: no tables (q > 2^20), quotients on the WIDE norm / Cramer forms
MASKED = [ga.GF(q) for q in (7, 2**8, 3**5, 3**10, 65521, 65537, 2**16, 2**32, 4294967291)]
WIDE = None


def run_ntt(entry, logn, batch):
    p, GF, F, adic = entry
    n = 1 << logn
    w = pow(GF._root_of_unity_int(n), int(rng.integers(0, max(1, n // 2))) * 2 + 1, p)
    x = rng.integers(0, p, (batch, n), dtype=np.uint32)
    x[int(rng.integers(0, batch))] = p - 1
    if batch > 1:
        x[1, ::2] = 0
        x[1, 1::2] = p - 1
    xt = torch.from_numpy(x.view(np.int32)).cuda()
    out = torch.empty_like(xt)
    L.check(lib.gfa_ntt(GF._handle, xt.data_ptr(), out.data_ptr(), n, batch, w, 0, L.U32, st))
    got = out.cpu().numpy().view(np.uint32)
    for i in {0, batch - 1, int(rng.integers(0, batch))}:
        assert np.array_equal(got[i], F.ntt_u32_pow2(x[i], w)), ("ntt vs oracle", p, logn, batch, i)
    if batch > 1:
        one = torch.empty_like(xt[:1])
        L.check(lib.gfa_ntt(GF._handle, xt[1:2].data_ptr(), one.data_ptr(), n, 1, w, 0, L.U32, st))
        assert torch.equal(one[0], out[1]), ("batched vs single", p, logn, batch)
    L.check(lib.gfa_ntt(GF._handle, out.data_ptr(), out.data_ptr(), n, batch, pow(w, p - 2, p), 1, L.U32, st))
    assert torch.equal(out, xt), ("ntt inverse", p, logn, batch)


t0 = time.time()
t_end = t0 + budget
next_large = t0 + 20.0
while time.time() < t_end:
    u = rng.random()
    if time.time() >= next_large:
        next_large = time.time() + 60.0
        entry = POOL[int(rng.integers(0, len(POOL)))]
        run_ntt(entry, int(rng.integers(13, min(entry[3], 20) + 1)), int(rng.integers(1, 3)))
        counts["ntt_large"] += 1
    elif u < 0.55:
        entry = POOL[int(rng.integers(0, len(POOL)))]
        logn = int(rng.integers(2, min(entry[3], 12) + 1))
        run_ntt(entry, logn, int(rng.choice([1, 2, 3, 5, 64, 70])) if logn <= 10 else int(rng.choice([1, 2, 5])))
        counts["ntt"] += 1
    elif u < 0.60:
        run_ntt(P16[int(rng.integers(0, len(P16)))], 16, int(rng.integers(64, 300)))
        counts["ntt16"] += 1
    elif u < 0.62:  # GF(65537), 2^10 .. 2^15 points, at least 2^22 in the batch: G = 2^16 / n transforms per workgroup (r06)
        logn = int(rng.integers(10, 16))
        run_ntt(POOL[0], logn, int(rng.integers(64, 130)) * (65536 >> logn) + int(rng.integers(0, 65536 >> logn)))
        counts["ntt_grouped"] += 1
    elif u < 0.66:
        q = int(EXT32_ORDERS[rng.integers(0, len(EXT32_ORDERS))])
        if q not in EXT32:
            try:
                G2 = ga.GF(q)
            except LookupError:  # no Conway polynomial in the shipped table: x^2 + x + c / x^3 + x + c, the first irreducible one
                pp, mm = next((r, e) for e in (2, 3) for r in (round(q ** (1 / e)),) if r**e == q)
                if mm == 2:
                    irr = [1, 1, next(c for c in range(1, pp) if pow((1 - 4 * c) % pp, (pp - 1) // 2, pp) == pp - 1)]
                else:
                    irr = [1, 0, 1, next(c for c in range(1, pp) if all((x * x * x + x + c) % pp for x in range(pp)))]
                G2 = ga.GF(q, irreducible_poly=irr)
            EXT32[q] = (G2, O.OracleField(G2.characteristic, G2.degree, int(G2.irreducible_poly), int(G2.primitive_element), lookup=q <= 2**20))
        G2, F2 = EXT32[q]
        n = int(rng.choice([1024, 1500, 4099, 20000, 70001]))
        off = int(rng.integers(0, 2)) * int(rng.integers(1, 5))
        a = rng.integers(0, q, n + off, dtype=np.uint64)
        b = rng.integers(1, q, n + off, dtype=np.uint64)
        a[rng.integers(0, n + off, 8)] = 0
        a[rng.integers(0, n + off, 8)] = q - 1
        b[rng.integers(0, n + off, 8)] = q - 1
        b[rng.integers(0, n + off, 8)] = G2.characteristic
        A, B = (G2(v.astype(np.uint32), dtype=np.uint32)[off:] for v in (a, b))
        a, b = a[off:], b[off:]
        u64 = lambda v: v.numpy().astype(np.uint64)
        k = int(rng.integers(0, n))
        full = lambda v: np.full(n, v, dtype=np.uint64)
        for got, want, what in ((A + B, F2.add(a, b), "add"), (A - B, F2.sub(a, b), "sub"), (-A, F2.neg(a), "neg"), (A * B, F2.mul(a, b), "mul"),
                                (A / B, F2.div(a, b), "div"), (np.reciprocal(B), F2.recip(b), "recip"),
                                (A / B[k], F2.div(a, full(b[k])), "array / scalar"), (A[k] / B, F2.div(full(a[k]), b), "scalar / array"),
                                (A * B[k], F2.mul(a, full(b[k])), "array * scalar"),
                                (B ** (kk := int(rng.integers(-q, q))), F2.pow(b, np.full(n, kk, dtype=np.int64)), "array ** scalar")):
            assert np.array_equal(u64(got), want), ("ext32", q, what, n, off)
        if 65536 < q <= 400000 and rng.random() < 0.15:  # arrays of at least 8 q elements: scalar powers through the per-call table of x ** k
            n2 = 8 * q + int(rng.integers(0, 5000))
            b2 = rng.integers(0, q, n2, dtype=np.uint64)
            kk = int(rng.integers(0, 2**40)) if rng.integers(0, 2) else int(rng.integers(-q, q))
            if kk < 0:
                b2[b2 == 0] = 1
            assert np.array_equal(u64(G2(b2.astype(np.uint32), dtype=np.uint32) ** kk), F2.pow(b2, np.full(n2, kk, dtype=np.int64))), ("ext32 pow table", q, kk, n2)
        if (a == 0).any():
            try:
                B / A
                raise AssertionError(("expected ZeroDivisionError", q))
            except ZeroDivisionError:
                pass
        counts["ext32"] += 1
    elif u < 0.675:
        q = int(LONG_ORDERS[rng.integers(0, len(LONG_ORDERS))])
        if q not in LONG:
            G3 = ga.GF(q)
            LONG[q] = (G3, O.OracleField(G3.characteristic, G3.degree, int(G3.irreducible_poly) if G3.degree > 1 else None, int(G3.primitive_element), lookup=q <= 2**16))
        G3, F3 = LONG[q]
        u64 = lambda v: v.numpy().astype(np.uint64)
        def tree(op, v):
            v = v.copy()
            while len(v) > 1:
                if len(v) & 1:
                    v = np.concatenate([op(v[:1], v[-1:]), v[1:-1]])
                h = len(v) // 2
                v = op(v[:h], v[h:])
            return int(v[0])
        n = int(rng.integers(70_000, 600_000))
        off = int(rng.integers(0, 2)) * int(rng.integers(1, 9))
        a = rng.integers(1, q, n + off, dtype=np.uint64)
        if rng.random() < 0.3:
            a[int(rng.integers(off, n + off))] = 0
        X = G3(a.astype(G3.dtypes[0]), dtype=G3.dtypes[0])[off:]
        a = a[off:]
        assert int(u64(np.add.reduce(X))) == tree(F3.add, a), ("long add.reduce", q, n, off)
        assert int(u64(np.multiply.reduce(X))) == tree(F3.mul, a), ("long multiply.reduce", q, n, off)
        for uf, op in ((np.add, F3.add), (np.multiply, F3.mul), (np.subtract, F3.sub)):
            out = u64(uf.accumulate(X))
            assert out[0] == a[0] and np.array_equal(op(out[:-1], a[1:]), out[1:]), ("long accumulate", uf.__name__, q, n, off)
        if G3.degree > 1:  # np.convolve through Karatsuba planes (na nb >= 2^20): c(x0) == a(x0) b(x0) at random points
            na, nb = int(rng.integers(1100, 5000)), int(rng.integers(1000, 3000))
            ca, cb = rng.integers(0, q, na, dtype=np.uint64), rng.integers(0, q, nb, dtype=np.uint64)
            cc = u64(np.convolve(G3(ca.astype(G3.dtypes[0]), dtype=G3.dtypes[0]), G3(cb.astype(G3.dtypes[0]), dtype=G3.dtypes[0])))
            x0 = rng.integers(0, q, 4, dtype=np.uint64)
            def horner(c):
                acc = np.zeros(4, dtype=np.uint64)
                for coef in c[::-1]:
                    acc = F3.add(F3.mul(acc, x0), np.full(4, coef, dtype=np.uint64))
                return acc
            assert len(cc) == na + nb - 1 and np.array_equal(horner(cc), F3.mul(horner(ca), horner(cb))), ("plane convolve", q, na, nb)
        counts["long"] += 1
    elif u < 0.72:
        p, GF, F, adic = POOL[int(rng.integers(0, len(POOL)))]
        na, nb = int(rng.integers(1500, 6000)), int(rng.integers(1500, 6000))
        a = rng.integers(0, p, na, dtype=np.uint64)
        b = rng.integers(0, p, nb, dtype=np.uint64)
        a[:2] = p - 1
        b[:2] = p - 1
        c = np.convolve(GF(a.astype(np.uint32)), GF(b.astype(np.uint32))).numpy().astype(np.uint64)
        assert c.shape == (na + nb - 1,)
        for k in [0, 1, na + nb - 2, int(rng.integers(0, na + nb - 1)), int(rng.integers(0, na + nb - 1))]:
            lo, hi = max(0, k - (nb - 1)), min(k, na - 1)
            want = int(np.add.reduce((a[lo:hi + 1].astype(object) * b[k - hi:k - lo + 1][::-1].astype(object)))) % p
            assert int(c[k]) == want, ("convolve", p, na, nb, k)
        counts["convolve"] += 1
    elif u < 0.96:
        GF = MASKED[int(rng.integers(0, len(MASKED)))]
        shape = (int(rng.integers(1, 5)), int(rng.integers(1, 200)))
        x = GF.Random(shape, seed=int(rng.integers(0, 2**31)))
        y = GF.Random(shape, low=1, seed=int(rng.integers(0, 2**31)))
        mask = rng.integers(0, 2, shape).astype(bool)
        wide = GF.dtypes[-1] if rng.random() < 0.5 else x.dtype  # the widest storage the field has (int64, or the only one)
        for uf in (np.add, np.subtract, np.multiply, np.true_divide):
            full = uf(x, y).numpy().astype(np.int64)
            old = GF(GF.Random(shape, seed=3).numpy(), dtype=wide)
            want = np.where(mask, full, old.numpy().astype(np.int64))
            got = uf(x, y, where=mask, out=old)
            assert got is old and np.array_equal(old.numpy().astype(np.int64), want), ("where", GF.name, uf.__name__, str(wide))
        init = GF.Random((), low=1, seed=int(rng.integers(0, 2**31)))
        axis = int(rng.integers(0, 2))
        got = np.add.reduce(x, axis=axis, where=mask, initial=init).numpy()
        want = (np.add.reduce(np.where(mask, x, GF.Zeros(shape)), axis=axis) + init).numpy()
        assert np.array_equal(got, want), ("reduce where", GF.name, axis)
        counts["where"] += 1
    else:
        if WIDE is None:
            import json
            WIDE = []
            for tag in ("GF_2e100", "GF_36893488147419103183", "GF_109987e4"):
                props = json.loads(str(np.load(os.path.join(ROOT, "tests", "golden", f"sage_wide_{tag}.npz"))["properties"]))
                pp, mm = props["characteristic"], props["degree"]
                WIDE.append(ga.GF(pp, primitive_element=props["primitive_element"]) if mm == 1 else
                            ga.GF(pp, mm, irreducible_poly=props["irreducible_poly"], primitive_element=props["primitive_element"]))
        import random
        GF = WIDE[int(rng.integers(0, 3))]
        q = GF.order
        n = int(rng.integers(2, 5))
        rnd = random.Random(int(rng.integers(0, 2**31)))
        A = GF(np.array([rnd.randrange(q) for _ in range(n * n)], dtype=object).reshape(n, n))
        B = GF(np.array([rnd.randrange(q) for _ in range(n * n)], dtype=object).reshape(n, n))
        try:
            assert np.array_equal(A @ np.linalg.inv(A), GF.Identity(n)), ("inv", GF.name)
        except np.linalg.LinAlgError:
            pass
        P, Lm, U = A.plu_decompose()
        assert np.array_equal(P @ Lm @ U, A), ("plu", GF.name)
        assert int(np.linalg.det(A @ B)) == int(np.linalg.det(A) * np.linalg.det(B)), ("det", GF.name)
        counts["wide"] += 1
el = time.time() - t0
total = sum(counts.values())
print(f"fuzz r06: {counts} = {total} cases in {el:.0f} s ({60.0 * total / el:.0f} per minute), seed {seed} -- identical to the oracle", flush=True)
