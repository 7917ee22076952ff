"""Pins oracle/gf_oracle.c against the reference itself, imported in place (build container only; skipped elsewhere)."""
import os
import sys
import warnings

import numpy as np
import pytest

from oracle import gf_oracle as O

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle", "ref_shim"))
import load_reference  # noqa: E402

pytestmark = pytest.mark.skipif(not load_reference.available(), reason="/root/reference is not present on this machine")


def _pair(order, lookup=False, **kw):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        GF = load_reference.ref_field(order, **kw)
    F = O.OracleField(GF.characteristic, GF.degree, int(GF.irreducible_poly) if GF.degree > 1 else None,
                      int(GF.primitive_element), lookup=lookup)
    return GF, F


def _eq(x, y):
    return [int(v) for v in np.asarray(x).ravel()] == [int(v) for v in np.asarray(y).ravel()]


@pytest.mark.parametrize("order,kw", [(2, {}), (4, {}), (2**8, {}), (5, {}), (31, {}), (3191, {}), (3**4, {}), (7**3, {}),
                                      (3**5, {}), (65537, {}), (2**8, dict(irreducible_poly=283, primitive_element=19)),
                                      (7**3, dict(irreducible_poly=643, primitive_element=244)), (2147483647, {}),
                                      (2**32, {}), (7340033, {}), (2**64 - 2**32 + 1, {}), (251**3, {})])
def test_elementwise_against_reference(order, kw):
    rng = np.random.default_rng(order % 9973)
    n = 120
    for lookup in ([False, True] if 2 < order <= 2**16 else [False]):
        GF, F = _pair(order, lookup, **kw)
        a = [int(rng.integers(0, 2**62)) * 3 % order for _ in range(n)]
        b = [int(rng.integers(0, 2**62)) * 5 % order for _ in range(n)]
        a[:3] = [0, 0, 0]
        b[1:4] = [0, 0, 0]
        ga, gb = GF(a), GF(b)
        oa, ob = np.array(a, dtype=object), np.array(b, dtype=object)
        assert _eq(F.add(oa, ob), ga + gb) and _eq(F.sub(oa, ob), ga - gb) and _eq(F.mul(oa, ob), ga * gb)
        assert _eq(F.neg(oa), -ga)
        bnz = [v or 1 for v in b]
        assert _eq(F.div(oa, np.array(bnz, dtype=object)), ga / GF(bnz))
        assert _eq(F.recip(np.array(bnz, dtype=object)), GF(bnz) ** -1)
        e = rng.integers(-20, 40, n)
        anz = [v or 1 for v in a]
        assert _eq(F.pow(np.array(anz, dtype=object), e), GF(anz) ** e)
        if lookup:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                GF.compile("jit-lookup")
                E, Lg, Z, ze = F.tables()
                assert np.array_equal(E, GF._EXP) and np.array_equal(Lg, GF._LOG) and np.array_equal(Z, GF._ZECH_LOG)
                assert ze == GF._ZECH_E
                GF.compile("python-calculate")


@pytest.mark.parametrize("order,n", [(769, 4), (65537, 256), (7340033, 64), (31, 30), (31, 15), (2**8, 255), (3**5, 22),
                                     (2**64 - 2**32 + 1, 32), (5**3, 124)])
def test_fft_against_reference(order, n):
    rng = np.random.default_rng(n)
    GF, F = _pair(order, lookup=order <= 2**16 and order != 31)
    x = [int(v) for v in rng.integers(0, min(order, 2**62), n)]
    assert _eq(F.ntt(x), np.fft.fft(GF(x)))
    assert _eq(F.ntt(x, inverse=True), np.fft.ifft(GF(x)))


@pytest.mark.parametrize("order,n,k,c,shorten", [(16, 15, 9, 1, 0), (16, 15, 9, 1, 3), (16, 15, 11, 3, 0), (81, 16, 10, 1, 0),
                                                 (27, 26, 20, 1, 0), (27, 13, 7, 1, 2), (256, 255, 223, 1, 0),
                                                 (31, 30, 22, 1, 0)])
def test_reed_solomon_against_reference(order, n, k, c, shorten):
    galois = load_reference.load()
    rng = np.random.default_rng(n * 100 + k)
    kw = dict(irreducible_poly=galois.matlab_primitive_poly(2, 8)) if order == 256 else {}
    GF, F = _pair(order, lookup=True, **kw)
    rs = galois.ReedSolomon(n, k, field=GF, c=c)
    R = O.OracleRS(F, n, k, c=c, alpha=int(rs.alpha))
    assert _eq(rs.generator_poly.coeffs, R.generator_poly) and _eq(rs.G, R.G) and _eq(rs.H, R.H)
    ks, ns, N = k - shorten, n - shorten, 6
    M = rng.integers(0, order, (N, ks))
    C = rs.encode(GF(M))
    assert _eq(C, R.encode(M))
    t = (n - k) // 2
    Rx = np.asarray(C).astype(np.int64).copy()
    E = np.zeros((N, ns), dtype=bool)
    for i in range(N):
        ne = min([0, t, t + 1, t // 2, 1, t + 3][i], ns)
        pos = rng.choice(ns, ne, replace=False)
        Rx[i, pos] = (Rx[i, pos] + rng.integers(1, order, ne)) % order
        if i >= 3:
            nu = min([2, n - k, n - k + 1][i % 3], ns)
            E[i, rng.choice(ns, nu, replace=False)] = True
    dref, nref = rs.decode(GF(Rx), erasures=E, output="codeword", errors=True)
    dmine, nmine = R.decode(Rx, erasures=E)
    assert np.array_equal(nref, nmine) and _eq(dref, dmine)
    assert np.array_equal(rs.detect(GF(Rx)), R.detect(Rx))


# ---- the "next" rows: BCH decoding, linear algebra, polynomial evaluation and logarithms, live against the reference ----
@pytest.mark.parametrize("p,n,k,d,c,shorten", [(2, 15, 7, None, 1, 0), (2, 31, None, 7, 1, 5), (2, 63, 45, None, 1, 0), (3, 26, 14, None, 1, 0),
                                                (3, 13, None, 5, 3, 2), (2, 15, None, 7, 3, 0)])
def test_bch_against_reference(p, n, k, d, c, shorten):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        galois = load_reference.load()
        GFp = load_reference.ref_field(p)
        m = galois.ilog(n, p) + 1
        ext = load_reference.ref_field(p**m, irreducible_poly=galois.matlab_primitive_poly(p, m))
        bch = galois.BCH(n, k, d, field=GFp, extension_field=ext, c=c)
    Fe = O.OracleField(p, m, int(ext.irreducible_poly), int(ext.primitive_element), lookup=True)
    B = O.OracleBCH(Fe, n, k, d=d, alpha=int(bch.alpha), c=c)
    assert (B.k, B.d) == (bch.k, bch.d)
    assert B.generator_poly == [int(v) for v in bch.generator_poly.coeffs]
    assert _eq(B.G, bch.G) and _eq(B.H, bch.H)
    rng = np.random.default_rng(n * 7 + c)
    ks, ns, t = bch.k - shorten, n - shorten, bch.t
    for trial in range(8):
        msg = rng.integers(0, p, ks)
        cw = np.asarray(bch.encode(GFp(msg))).astype(np.int64)
        assert _eq(B.encode(msg)[0], cw)
        r = cw.copy()
        ne = int(rng.integers(0, t + 2))
        pos = rng.choice(ns, min(ne, ns), replace=False)
        r[pos] = (r[pos] + rng.integers(1, p, pos.size)) % p
        er = np.zeros(ns, dtype=bool)
        if trial % 3 == 2:
            er[rng.choice(ns, int(rng.integers(0, bch.d)), replace=False)] = True
        odec, onerr = B.decode(r, er if er.any() else None)
        try:
            dec, nerr = bch.decode(GFp(r), erasures=er if er.any() else None, output="codeword", errors=True)
        except (ValueError, OverflowError):
            assert ((odec < 0) | (odec >= p)).any()  # the reference rejects symbols outside GF(p)
            continue
        assert int(nerr) == int(onerr[0]) and _eq(dec, odec[0])
        assert bool(bch.detect(GFp(r))) == bool(B.detect(r)[0])


@pytest.mark.parametrize("order", [2, 5, 31, 2**8, 3**3, 65537, 2147483647, 2**64 - 2**32 + 1])
def test_linear_algebra_against_reference(order):
    GF, F = _pair(order)
    rng = np.random.default_rng(order % 1009)

    def rnd(shape):
        return np.array([int(rng.integers(0, 2**62)) % order for _ in range(int(np.prod(shape)))], dtype=object).reshape(shape)

    for m, n in [(1, 1), (3, 3), (4, 6), (6, 4), (5, 5)]:
        A = rnd((m, n))
        if m == 5:
            A[2] = A[0]  # rank-deficient
            A[:, 1] = 0
        gA = GF(A)
        assert _eq(F.row_reduce(A)[0], gA.row_reduce())
        P, Lm, U, _ = F.plu_decompose(A)
        rp, rl, ru = gA.plu_decompose()
        assert _eq(P, rp) and _eq(Lm, rl) and _eq(U, ru)
        B = rnd((n, 3))
        assert _eq(F.matmul(A, B), gA @ GF(B))
        assert F.matrix_rank(A) == np.linalg.matrix_rank(gA)
        for op in ("row_space", "column_space", "left_null_space", "null_space"):
            want = getattr(gA, op)()
            got = getattr(F, op)(A)
            assert want.size == got.size and _eq(got, want), op
        if m == n:
            assert F.det(A) == int(np.linalg.det(gA))
            try:
                inv = np.linalg.inv(gA)
            except np.linalg.LinAlgError:
                with pytest.raises(np.linalg.LinAlgError):
                    F.inv(A)
            else:
                assert _eq(F.inv(A), inv)
                b = rnd((n,))
                assert _eq(F.solve(A, b), np.linalg.solve(gA, GF(b)))
            try:
                rl2, ru2 = gA.lu_decompose()
            except ValueError:
                with pytest.raises(ValueError):
                    F.lu_decompose(A)
            else:
                l2, u2 = F.lu_decompose(A)
                assert _eq(l2, rl2) and _eq(u2, ru2)


@pytest.mark.parametrize("order", [31, 2**8, 3**4, 65537])
def test_poly_evaluate_and_log_against_reference(order):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        galois = load_reference.load()
    GF, F = _pair(order, lookup=True)
    rng = np.random.default_rng(order)
    coeffs = rng.integers(0, order, 9)
    coeffs[0] = 1
    x = rng.integers(0, order, 40)
    assert _eq(F.poly_eval(coeffs, x), galois.Poly(coeffs, field=GF)(GF(x)))
    xs = rng.integers(1, order, 40)
    _, LOG, _, _ = F.tables()
    assert _eq(LOG[xs], np.log(GF(xs)))


def test_committed_reference_vectors_regenerate_identically(tmp_path):
    """tests/golden/generate_golden.py (reference, reference_bch, reference_wide, reference_tables) run against /root/reference must reproduce
    the committed .npz files array for array: the fixtures are outputs of the reference, not of this repository."""
    import subprocess
    import sys

    gen = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "generate_golden.py")
    env = dict(os.environ, GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, gen, "reference", "reference_bch", "reference_wide", "reference_tables"], capture_output=True, text=True,
                       env=env, cwd=os.path.dirname(gen), timeout=1800)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    for name in ("reference_outputs.npz", "reference_bch_outputs.npz", "reference_wide_codes.npz", "reference_table_fields.npz"):
        new = np.load(os.path.join(str(tmp_path), name), allow_pickle=True)
        old = np.load(os.path.join(os.path.dirname(gen), name), allow_pickle=True)
        assert sorted(new.files) == sorted(old.files), name
        for k in new.files:
            assert np.array_equal(new[k], old[k]), (name, k)


FACTORY_ORDERS = [2, 3, 5, 7, 31, 251, 257, 509, 3191, 8191, 32749, 65521, 65537, 7340033, 2147483647, 4294967291, 2**61 - 1,
                  2**64 - 2**32 + 1, 18446744073709551557, 2**2, 2**3, 2**4, 2**8, 2**9, 2**10, 2**12, 2**13, 2**14, 2**15, 2**16, 2**17,
                  2**20, 2**24, 2**31, 2**32, 3**2, 3**4, 3**5, 3**7, 3**9, 3**10, 5**3, 5**5, 7**3, 7**5, 11**4, 13**4, 127**2, 251**2,
                  251**3, 31**5, 2**100, 36893488147419103183, 109987**4]


@pytest.mark.parametrize("order", FACTORY_ORDERS)
def test_field_factory_matches_the_reference(order):
    """galois_amd.GF(order) against galois.GF(order) of the reference, live: name, characteristic / degree / order, default
    irreducible polynomial (Conway table), default primitive element, is_primitive_poly, the dtype list and the array-free parts
    of `properties` -- the host side of SURVEY.md 8(a1) / (a3) for every size class the device kernels distinguish."""
    import galois_amd as ga

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        R = load_reference.ref_field(order)
    G = ga.GF(order)
    assert (G.name, G.characteristic, G.degree, G.order) == (R.name, R.characteristic, R.degree, R.order)
    assert int(G.irreducible_poly) == int(R.irreducible_poly)
    assert int(G.primitive_element) == int(R.primitive_element)
    assert bool(G.is_primitive_poly) == bool(R.is_primitive_poly)
    assert G.is_prime_field == R.is_prime_field and G.is_extension_field == R.is_extension_field
    assert [np.dtype(d) for d in G.dtypes] == [np.dtype(d) for d in R.dtypes]
    assert str(G.irreducible_poly) == str(R.irreducible_poly)


def test_number_theory_matches_the_reference():
    """galois_amd/_numtheory.py (what the field factory and galois.ntt's modulus choice stand on) against the reference's functions
    of the same names, live: primality, factorisation, primitive roots, irreducibility / primitivity of polynomials, the MATLAB
    default polynomials and the Conway table."""
    from galois_amd import _numtheory as nt

    galois = load_reference.load()
    rng = np.random.default_rng(7)
    numbers = [1, 2, 3, 4, 31, 255, 256, 257, 65535, 65536, 65537, 7340033, 2**31 - 1, 2**32 - 1, 2**32 + 1, 4294967291, 2**61 - 1,
               2**64 - 2**32 + 1, 2**64 - 2**32, 3**20, 2**63, 1000003 * 999983] + [int(v) for v in rng.integers(2, 2**40, 40)]
    for n in numbers:
        assert nt.is_prime(n) == bool(galois.is_prime(n)), n
        if n > 1:
            f, e = galois.factors(n)
            assert nt.factors(n) == ([int(v) for v in f], [int(v) for v in e]), n
    for p in (2, 3, 5, 7, 31, 257, 769, 7681, 12289, 65537, 7340033, 469762049, 2**31 - 1, 2**61 - 1, 2**64 - 2**32 + 1):
        assert nt.primitive_root(p) == int(galois.primitive_root(p)), p
    for p, m in ((2, 2), (2, 8), (2, 16), (2, 32), (3, 5), (3, 10), (5, 4), (7, 3), (127, 2), (251, 3), (2, 100), (109987, 4)):
        assert nt.conway_poly(p, m) == int(galois.conway_poly(p, m)), (p, m)
    for p, m in ((2, 2), (2, 3), (2, 4), (2, 7), (2, 8), (2, 16)):  # (odd p: the reference's polynomial search needs its JIT arrays)
        assert nt.matlab_primitive_poly(p, m) == int(galois.matlab_primitive_poly(p, m)), (p, m)
    for val in (0x11B, 0x11D, 0x11C, 0x1002D, 0x10001, 0b111, 0b1011, 0b1111, 0x100008299):  # over GF(2)
        coeffs = nt.poly_from_int(val, 2)
        ref = galois.Poly.Int(val)
        assert nt.is_irreducible(coeffs, 2) == bool(ref.is_irreducible()), hex(val)
        if nt.is_irreducible(coeffs, 2):
            assert nt.is_primitive_poly(coeffs, 2) == bool(ref.is_primitive()), hex(val)


@pytest.mark.parametrize("x,size", [([1, 2, 3, 4], None), ([1, 2, 3, 4], 8), ([0, 0, 0, 0], None), ([100, 200, 300], 4), ([65536] * 16, None),
                                    ([12288] * 8, 1024), ([7340032, 5], 1 << 20), (list(range(30)), 32), ([2**31, 1, 2, 3], None)])
def test_default_ntt_modulus_matches_the_reference(x, size):
    """galois.ntt(x, size) without a modulus: the field the reference's result lives in against galois_amd._ntt._default_modulus."""
    from galois_amd._ntt import _default_modulus

    galois = load_reference.load()
    n = size or len(x)
    want = _default_modulus(max(x), n)
    if n <= 64:  # run the reference's transform itself for the small cases
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            X = galois.ntt(x, size=size)
        assert int(type(X).characteristic) == want
    m = int(np.ceil(max(x) / n))
    while not galois.is_prime(m * n + 1):
        m += 1
    assert want == m * n + 1
