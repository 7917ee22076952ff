"""Differential fuzzing of the round-5 paths against the oracle.  Usage: python tools/fuzz_r05.py [seconds] [seed]
  * signed-Montgomery NTT kernels over RANDOM NTT primes below 2^29 (every BMAX class, gfa_m32_net.h): random power-of-two lengths
    2^5 .. 2^22 the prime's 2-adicity allows (one-pass, one-workgroup, two-pass and three-pass forms), random primitive roots, rows at
    the magnitude limit; forward against the oracle, scaled inverse as a round trip, batches against the same rows one by one;
  * np.convolve over random primes below 2^32 at lengths that take the CRT route (either prime set) against the direct kernel
    (short operands) and, on a sample of coefficients, against Python integers;
  * `where=` / `initial=` on random ufunc calls and reductions of random fields against the unmasked result blended by NumPy;
  * two-limb fields (GF(2^100), GF(36893488147419103183), GF(109987^4)): random systems -- A @ inv(A) = I, P L U = A, det(A B) =
    det(A) det(B), solve, alpha ** log(x) = x, ifft(fft(x)) = x, sqrt(x * x) ** 2 = x * x."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd._ntt import fft_batched
from oracle import gf_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(seed)
t_end = time.time() + budget
counts = {"ntt": 0, "convolve": 0, "where": 0, "wide": 0}


def ntt_prime():
    while True:
        bits = int(rng.integers(20, 30))
        adic = int(rng.integers(8, min(bits - 1, 23)))
        c = int(rng.integers(1 << (bits - adic - 1), 1 << (bits - adic))) | 1
        p = c * (1 << adic) + 1
        if p < 2**29 and ga.is_prime(p):
            return p, adic


def _wide_fields():
    import json
    out = []
    for tag in ("GF_2e100", "GF_36893488147419103183", "GF_109987e4"):  # the reference's three big Sage folders: their own polynomials
        props = json.loads(str(np.load(os.path.join(ROOT, "tests", "golden", f"sage_wide_{tag}.npz"))["properties"]))
        pp, mm = props["characteristic"], props["degree"]
        out.append(ga.GF(pp, primitive_element=props["primitive_element"]) if mm == 1 else
                   ga.GF(pp, mm, irreducible_poly=props["irreducible_poly"], primitive_element=props["primitive_element"]))
    return out


WIDE = _wide_fields()
ints = lambda a: np.array([int(v) for v in np.asarray(a.numpy()).ravel()], dtype=object).reshape(a.shape)

while time.time() < t_end:
    kind = rng.integers(0, 4)
    if kind == 0:
        p, adic = ntt_prime()
        GF = ga.GF(p)
        F = O.OracleField(p, 1, None, int(GF.primitive_element))
        logn = int(rng.integers(5, min(adic, 22) + 1))
        n = 1 << logn
        batch = int(rng.integers(1, 5)) if logn > 16 else int(rng.choice([1, 2, 5, 64, 70]))
        if n * batch > (1 << 24):
            batch = 1
        w = pow(GF._root_of_unity_int(n), int(rng.integers(0, n // 2)) * 2 + 1, p)
        x = rng.integers(0, p, (batch, n), dtype=np.uint32)
        x[int(rng.integers(0, batch))] = p - 1
        if batch > 1:
            x[1, ::2] = 0; x[1, 1::2] = p - 1
        from galois_amd import _lib as L
        lib = L.lib(); st = torch.cuda.current_stream().cuda_stream
        xt = torch.from_numpy(x.view(np.int32)).cuda(); out = torch.empty_like(xt)
        L.check(lib.gfa_ntt(GF._handle, xt.data_ptr(), out.data_ptr(), n, batch, w, 0, L.U32, st))
        got = out.cpu().numpy().view(np.uint32)
        for i in set([0, batch - 1, int(rng.integers(0, batch))]):
            assert np.array_equal(got[i], F.ntt_u32_pow2(x[i], w)), ("ntt vs oracle", p, logn, batch, i)
        if batch > 1:
            one = torch.empty_like(xt[:1])
            L.check(lib.gfa_ntt(GF._handle, xt[1:2].data_ptr(), one.data_ptr(), n, 1, w, 0, L.U32, st))
            assert torch.equal(one[0], out[1]), ("batched vs single", p, logn, batch)
        L.check(lib.gfa_ntt(GF._handle, out.data_ptr(), out.data_ptr(), n, batch, pow(w, p - 2, p), 1, L.U32, st))
        assert torch.equal(out, xt), ("ntt inverse", p, logn, batch)
        counts["ntt"] += 1
    elif kind == 1:
        while True:
            p = int(rng.integers(3, 2**32)) | 1
            if ga.is_prime(p):
                break
        GF = ga.GF(p)
        na, nb = int(rng.integers(2000, 60000)), int(rng.integers(2000, 60000))
        a = rng.integers(0, p, na, dtype=np.uint64); b = rng.integers(0, p, nb, dtype=np.uint64)
        a[:2] = p - 1; b[:2] = p - 1
        c = np.convolve(GF(a.astype(np.uint32)), GF(b.astype(np.uint32))).numpy().astype(np.uint64)
        assert c.shape == (na + nb - 1,)
        for k in [0, 1, na + nb - 2, int(rng.integers(0, na + nb - 1)), int(rng.integers(0, na + nb - 1))]:
            lo, hi = max(0, k - (nb - 1)), min(k, na - 1)
            want = sum(int(a[i]) * int(b[k - i]) for i in range(lo, hi + 1)) % p
            assert int(c[k]) == want, ("convolve", p, na, nb, k)
        counts["convolve"] += 1
    elif kind == 2:
        order = int(rng.choice([7, 2**8, 3**5, 65537, 2**16, 2**32, 4294967291, 2**61 - 1]))
        GF = ga.GF(order)
        shape = (int(rng.integers(1, 6)), int(rng.integers(1, 300)))
        x = GF.Random(shape, seed=int(rng.integers(0, 2**31))); y = GF.Random(shape, low=1, seed=int(rng.integers(0, 2**31)))
        mask = rng.integers(0, 2, shape).astype(bool)
        for uf in (np.add, np.subtract, np.multiply, np.true_divide):
            full = ints(uf(x, y))
            old = GF.Random(shape, seed=3)
            want = np.where(mask, full, ints(old))
            assert np.array_equal(ints(uf(x, y, where=mask, out=old)), want), ("where", order, uf.__name__)
        init = GF.Random((), low=1, seed=int(rng.integers(0, 2**31)))
        axis = int(rng.integers(0, 2))
        got = ints(np.add.reduce(x, axis=axis, where=mask, initial=init))
        # masked sum = sum of (x where mask else 0) + init, by the unmasked kernels
        zero = GF.Zeros(shape)
        want = ints(np.add.reduce(np.where(mask, x, zero), axis=axis) + init)
        assert np.array_equal(got, want), ("reduce where", order, axis)
        counts["where"] += 1
    else:
        GF = WIDE[int(rng.integers(0, 3))]
        q = GF.order
        n = int(rng.integers(2, 6))
        import random
        rnd = random.Random(int(rng.integers(0, 2**31)))
        A = GF(np.array([rnd.randrange(q) for _ in range(n * n)], dtype=object).reshape(n, n))
        B = GF(np.array([rnd.randrange(q) for _ in range(n * n)], dtype=object).reshape(n, n))
        I = GF.Identity(n)
        try:
            Ai = np.linalg.inv(A)
            assert np.array_equal(A @ Ai, I), ("inv", GF.name)
            v = GF(np.array([rnd.randrange(q) for _ in range(n)], dtype=object))
            assert np.array_equal(A @ np.linalg.solve(A, v), v), ("solve", GF.name)
        except np.linalg.LinAlgError:
            pass
        P, Lm, U = A.plu_decompose()
        assert np.array_equal(P @ Lm @ U, A), ("plu", GF.name)
        assert int(np.linalg.det(A @ B)) == int(np.linalg.det(A) * np.linalg.det(B)), ("det", GF.name)
        x = GF(np.array([rnd.randrange(1, q) for _ in range(4)], dtype=object))
        lg = np.log(x)
        alpha = GF(np.array(GF._primitive_element_int, dtype=object))
        assert np.array_equal(alpha ** lg, x), ("log", GF.name)
        divs = [d for d in (2, 3, 4, 5, 6, 8, 9, 12, 16, 24, 33) if (q - 1) % d == 0]
        d = int(rng.choice(divs))
        z = GF(np.array([rnd.randrange(q) for _ in range(d)], dtype=object))
        assert np.array_equal(np.fft.ifft(np.fft.fft(z)), z), ("fft", GF.name, d)
        s = np.sqrt(x * x)
        assert np.array_equal(s * s, x * x), ("sqrt", GF.name)
        counts["wide"] += 1
print("fuzz r05:", counts, "seed", seed, "-- identical to the oracle", flush=True)
