"""Differential fuzzing of the Reed-Solomon / BCH device paths against the oracle: random codes (field, n, k or d, c,
systematic or not), random shortening, 0 .. t+3 errors and 0 .. d erasures per word.  Every decoded word, error count,
detect flag and encoder output must match bit for bit (rows on which the reference would raise are checked to raise).
Usage: python tools/fuzz_codes.py [seconds] [seed] [wide]   (wide: codes whose syndrome field has 512 .. 4096 elements)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import galois_amd as ga
from oracle import gf_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
WIDE = len(sys.argv) > 3 and sys.argv[3] == "wide"
rng = np.random.default_rng(seed)
t_end = time.time() + budget
n_codes = n_words = 0
fields = {}


def field(q):
    if q not in fields:
        GF = ga.GF(q)
        fields[q] = (GF, O.OracleField(GF.characteristic, GF.degree, int(GF.irreducible_poly) if GF.degree > 1 else None,
                                      int(GF.primitive_element), lookup=True))
    return fields[q]


while time.time() < t_end:
    kind = rng.choice(["rs", "rs", "bch"])
    if kind == "rs":
        q = int(rng.choice([512, 1024, 1024, 2048, 4096, 729, 343, 625, 2401, 257, 1009]) if WIDE else
                rng.choice([8, 16, 32, 64, 128, 256, 256, 256, 27, 81, 125, 31, 251]))
        GF, F = field(q)
        divs = [d for d in range(3, q) if (q - 1) % d == 0]
        n = int(rng.choice(divs + [q - 1] * 3))
        k = int(rng.integers(1, n + 1))
        if WIDE:
            k = n - int(rng.integers(0, min(n - 1, 120) + 1))
        elif rng.random() < 0.5 and n > 8:  # favour parity lengths that take the LFSR / wave-kernel fast path
            nk = int(rng.choice([x for x in range(4, min(n - 1, 60) + 1, 4)] or [n - k]))
            k = n - nk
        c = int(rng.choice([0, 1, 1, 2, 5]))
        sysm = bool(rng.random() < 0.8)
        code = ga.ReedSolomon(n, k, field=GF, c=c, systematic=sysm)
        oc = O.OracleRS(F, n, k, alpha=code.alpha, c=c) if sysm else None
        p = GF.order
        base_p = 0
    else:
        p = int(rng.choice([2, 2, 2, 3, 5]))
        m = int(rng.integers({2: 9, 3: 6, 5: 4}[p], {2: 13, 3: 8, 5: 6}[p])) if WIDE else int(rng.integers(2, {2: 9, 3: 6, 5: 4}[p]))
        if (p**m > 256) != WIDE:
            continue
        n = p**m - 1
        dsg = int(rng.integers(2, min(n, 40)))
        c = int(rng.choice([0, 1, 1, 3]))
        sysm = bool(rng.random() < 0.8)
        try:
            code = ga.BCH(n, d=dsg, field=ga.GF(p), c=c, systematic=sysm)
        except ValueError:
            continue
        ext = code.extension_field
        Fe = O.OracleField(p, ext.degree, int(ext.irreducible_poly), int(ext.primitive_element), lookup=True)
        oc = O.OracleBCH(Fe, n, d=dsg, alpha=code.alpha, c=c, systematic=sysm)
        k = code.k
        base_p = p
    d = code.d
    t = (d - 1) // 2
    shorten = int(rng.integers(0, k)) if rng.random() < 0.4 else 0
    ks, ns = k - shorten, n - shorten
    N = int(rng.integers(1, 40 if WIDE else 300))
    M = rng.integers(0, p, (N, ks))
    C = code.encode(M).numpy().astype(np.int64)
    if oc is not None:
        assert np.array_equal(C, np.asarray(oc.encode(M)).astype(np.int64)), ("encode", kind, n, k, c, sysm, shorten)
    R = C.copy()
    E = np.zeros((N, ns), dtype=bool)
    for i in range(N):
        ne = int(rng.integers(0, min(ns, t + 3) + 1))
        pos = rng.choice(ns, ne, replace=False)
        R[i, pos] = (R[i, pos] + rng.integers(1, p, ne)) % p
        if rng.random() < 0.4:
            E[i, rng.choice(ns, int(rng.integers(0, min(ns, d) + 1)), replace=False)] = True
    use_eras = bool(E.any()) and rng.random() < 0.8
    if oc is None:  # non-systematic RS: round-trip properties only
        ok = np.array([(R[i] != C[i]).sum() <= t for i in range(N)])
        dec, nerr = code.decode(R[ok], errors=True) if ok.any() else (None, None)
        if ok.any():
            assert np.array_equal(dec.numpy().astype(np.int64), M[ok]), ("nonsys decode", n, k, c)
    else:
        odec, onerr = oc.decode(R, E if use_eras else None)
        odec = np.asarray(odec).astype(np.int64)
        bad = ((odec < 0) | (odec >= p)).any(axis=1) if base_p else np.zeros(N, dtype=bool)
        okr = ~bad
        if okr.any():
            dec, nerr = code.decode(R[okr], erasures=E[okr] if use_eras else None, output="codeword", errors=True)
            assert np.array_equal(np.atleast_1d(nerr), onerr[okr]), ("n_errors", kind, n, k, d, c, sysm, shorten, use_eras)
            assert np.array_equal(dec.numpy().astype(np.int64).reshape(-1, ns), odec[okr]), ("decoded", kind, n, k, d, c, sysm, shorten, use_eras)
        for i in np.nonzero(bad)[0][:3]:
            try:
                code.decode(R[i], erasures=E[i] if use_eras else None)
                raise AssertionError(("expected ValueError", kind, n, k, d, c))
            except ValueError:
                pass
        assert np.array_equal(np.atleast_1d(code.detect(R)), np.atleast_1d(oc.detect(R))), ("detect", kind, n, k, c, shorten)
    n_codes += 1
    n_words += N
print(f"fuzz_codes: {n_codes} codes, {n_words} words, all outputs identical to the oracle (seed {seed}, {budget:.0f} s)")
