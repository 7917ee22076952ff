"""Differential fuzzing of transforms and linear algebra against the oracle: NTT / inverse NTT of random lengths (powers
of two up to 2^16 and mixed radix) over random fields incl. extension fields; matmul, row reduction, PLU, inverse,
determinant on random (often rank-deficient) matrices.  Usage: python tools/fuzz_ntt_linalg.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import galois_amd as ga
from oracle import gf_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(seed)
NTT_FIELDS = [65537, 7340033, 13 * 2**20 + 1, 998244353, 3221225473, 2**64 - 2**32 + 1, 769, 31, 2**8, 3**5, 2**12, 257, 12289]
LA_FIELDS = [2, 3, 31, 251, 2**8, 65537, 2**31 - 1, 3**5, 2**16, 2**64 - 2**32 + 1, 2**4, 7**3]
cache = {}


def pair(q):
    if q not in cache:
        GF = ga.GF(q)
        cache[q] = (GF, O.OracleField(GF.characteristic, GF.degree, int(GF.irreducible_poly) if GF.degree > 1 else None,
                                      int(GF.primitive_element), lookup=q <= 2**16))
    return cache[q]


def rnd(q, shape):
    if q > 2**63:
        return (rng.integers(0, 2**63, shape, dtype=np.uint64) * 2 + rng.integers(0, 2, shape, dtype=np.uint64)) % np.uint64(q)
    return rng.integers(0, q, shape, dtype=np.uint64)


def u64(a):
    a = np.asarray(a)
    return np.array([int(x) for x in a.ravel()], dtype=np.uint64).reshape(a.shape) if a.dtype == object else a.astype(np.uint64)


t_end = time.time() + budget
n_ntt = n_la = 0
while time.time() < t_end:
    if rng.random() < 0.5:
        q = int(NTT_FIELDS[rng.integers(0, len(NTT_FIELDS))])
        GF, F = pair(q)
        divs = [d for d in range(2, min(q, 70000)) if (q - 1) % d == 0 and d <= 65536]
        pow2 = [d for d in divs if d & (d - 1) == 0]
        n = int(rng.choice(pow2)) if (pow2 and rng.random() < 0.7) else int(rng.choice(divs))
        if n > 4096 and q > 2**63:
            n = 4096
        x = rnd(q, n)
        gx = GF([int(v) for v in x]) if q > 2**63 else GF(x)
        X = np.fft.fft(gx)
        assert np.array_equal(u64(X.numpy()), F.ntt(x)), ("fft", q, n)
        assert np.array_equal(u64(np.fft.ifft(X).numpy()), x), ("ifft(fft)", q, n)
        assert np.array_equal(u64(np.fft.ifft(gx).numpy()), F.ntt(x, inverse=True)), ("ifft", q, n)
        n_ntt += 1
    else:
        q = int(LA_FIELDS[rng.integers(0, len(LA_FIELDS))])
        GF, F = pair(q)
        if q in (2, 3, 31, 251) and rng.random() < 0.3:  # large enough for the matrix-core path (centred int8 residues)
            m, k, n = (int(v) for v in rng.integers(100, 420, 3))
            A, B = rnd(q, (m, k)), rnd(q, (k, n))
            assert np.array_equal(u64((GF(A) @ GF(B)).numpy()), F.matmul(A, B)), ("matmul mfma", q, m, k, n)
            n_la += 1
            continue
        if rng.random() < 0.06:  # r06: GF(2^m) on the matrix cores as Karatsuba bit planes (M, N >= 128, at least 2^24 multiply-adds)
            mb = int(rng.integers(2, 17))
            GFb, Fb = pair(2**mb)
            m, k, n = 128 + int(rng.integers(0, 300)), 1024 + int(rng.integers(0, 300)), 128 + int(rng.integers(0, 300))
            A, B = rnd(2**mb, (m, k)), rnd(2**mb, (k, n))
            assert np.array_equal(u64((GFb(A) @ GFb(B)).numpy()), Fb.matmul(A, B)), ("matmul bit planes", mb, m, k, n)
            n_la += 1
            continue
        if rng.random() < 0.06:  # r06: GF(p^m), odd p <= 251, on the matrix cores as Karatsuba digit planes
            qe = int([3**2, 3**3, 3**5, 5**3, 7**3, 3**7, 11**2, 251**2, 13**3, 5**5, 3**9][rng.integers(0, 11)])
            GFe, Fe = pair(qe)
            m, k, n = 128 + int(rng.integers(0, 200)), 1024 + int(rng.integers(0, 300)), 128 + int(rng.integers(0, 200))
            A, B = rnd(qe, (m, k)), rnd(qe, (k, n))
            assert np.array_equal(u64((GFe(A) @ GFe(B)).numpy()), Fe.matmul(A, B)), ("matmul digit planes", qe, m, k, n)
            n_la += 1
            continue
        if q in (65537, 2**31 - 1) and rng.random() < 0.05:  # 7-bit limb path (>= 2^27 multiply-adds)
            m, k, n = 512 + int(rng.integers(0, 40)), 512 + int(rng.integers(0, 40)), 512 + int(rng.integers(0, 40))
            A, B = rnd(q, (m, k)), rnd(q, (k, n))
            assert np.array_equal(u64((GF(A) @ GF(B)).numpy()), F.matmul(A, B)), ("matmul limbs", q, m, k, n)
            n_la += 1
            continue
        m, n, k = (int(v) for v in rng.integers(1, 40, 3))
        A, B = rnd(q, (m, k)), rnd(q, (k, n))
        mk = (lambda a: GF([[int(v) for v in r] for r in a])) if q > 2**63 else (lambda a: GF(a))
        assert np.array_equal(u64((mk(A) @ mk(B)).numpy()), F.matmul(A, B)), ("matmul", q, m, k, n)
        S = rnd(q, (m, m))
        if rng.random() < 0.4 and m > 2:
            r = int(rng.integers(1, m))
            S = np.asarray(F.matmul(rnd(q, (m, r)), rnd(q, (r, m))))
        gS = mk(S)
        assert np.array_equal(u64(gS.row_reduce().numpy()), F.row_reduce(S)[0]), ("row_reduce", q, m)
        P_, L_, U_, _ = F.plu_decompose(S)
        p, l, u = gS.plu_decompose()
        assert np.array_equal(u64(p.numpy()), P_) and np.array_equal(u64(l.numpy()), L_) and np.array_equal(u64(u.numpy()), U_), ("plu", q, m)
        assert int(np.linalg.det(gS)) == F.det(S), ("det", q, m)
        try:
            want = F.inv(S)
        except np.linalg.LinAlgError:
            try:
                np.linalg.inv(gS)
                raise AssertionError(("expected LinAlgError", q, m))
            except np.linalg.LinAlgError:
                pass
        else:
            assert np.array_equal(u64(np.linalg.inv(gS).numpy()), want), ("inv", q, m)
        R = rnd(q, (m, n))
        assert np.array_equal(u64(mk(R).row_reduce().numpy()), F.row_reduce(R)[0]), ("row_reduce rect", q, m, n)
        n_la += 1
print(f"fuzz_ntt_linalg: {n_ntt} transforms and {n_la} matrix cases, every result identical to the oracle (seed {seed}, {budget:.0f} s)")
