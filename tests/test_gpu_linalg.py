"""Parity of the field linear-algebra kernels against the reference's Sage fixtures
(tests/fields/test_linalg.py) and the oracle.  Bit-exact, including the pivot order of L, U and P."""
import numpy as np
import pytest

import galois_amd as ga
from oracle import gf_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _field(props):
    p, m = props["characteristic"], props["degree"]
    if m == 1:
        return ga.GF(p, primitive_element=int(props["primitive_element"]))
    return ga.GF(p, m, irreducible_poly=H.poly_coeffs_to_int(props["irreducible_poly"], p),
                 primitive_element=int(props["primitive_element"]))


@pytest.mark.parametrize("tag", H.SAGE_LINALG)
def test_sage_linalg_fixtures(tag):
    props, d = H.load_sage_linalg(tag)
    GF = _field(props)
    rng = np.random.default_rng(5)

    def arr(x):
        dt = GF.dtypes[int(rng.integers(0, len(GF.dtypes)))]  # the reference tests draw a random legal dtype too
        return GF(x.astype(np.uint64).astype(dt) if x.dtype != object else x, dtype=dt)

    for X, Y, Z in H.linalg_cases(d, "matrix_multiply", "XYZ"):
        z = arr(X) @ arr(Y)
        assert type(z) is GF
        H.assert_equal_ints(z.numpy(), Z, "matmul")
    for X, Z in H.linalg_cases(d, "row_reduce", "XZ"):
        H.assert_equal_ints(arr(X).row_reduce().numpy(), Z, "row_reduce")
    for X, Lt, Ut in H.linalg_cases(d, "lu_decompose", "XLU"):
        l, u = arr(X).lu_decompose()
        H.assert_equal_ints(l.numpy(), Lt, "lu L")
        H.assert_equal_ints(u.numpy(), Ut, "lu U")
    for X, Pt, Lt, Ut in H.linalg_cases(d, "plu_decompose", "XPLU"):
        p, l, u = arr(X).plu_decompose()
        H.assert_equal_ints(p.numpy(), Pt, "plu P")
        H.assert_equal_ints(l.numpy(), Lt, "plu L")
        H.assert_equal_ints(u.numpy(), Ut, "plu U")
        H.assert_equal_ints((p @ l @ u).numpy(), X, "P L U = A")
    for X, Z in H.linalg_cases(d, "matrix_inverse", "XZ"):
        H.assert_equal_ints(np.linalg.inv(arr(X)).numpy(), Z, "inv")
    for X, Z in H.linalg_cases(d, "matrix_determinant", "XZ"):
        assert int(np.linalg.det(arr(X))) == int(Z), "det"
    for X, Y, Z in H.linalg_cases(d, "matrix_solve", "XYZ"):
        H.assert_equal_ints(np.linalg.solve(arr(X), arr(Y)).numpy(), Z, "solve")
    for op in ("row_space", "column_space", "left_null_space", "null_space"):
        for X, Z in H.linalg_cases(d, op, "XZ"):
            got = getattr(arr(X), op)().numpy()
            if Z.size == 0:
                assert got.size == 0, op
            else:
                H.assert_equal_ints(got, Z.reshape(got.shape), op)


_FIELDS = [("gf256", 2, 8, 285, 2), ("gf2", 2, 1, None, 1), ("gf31", 31, 1, None, 3), ("gf65537", 65537, 1, None, 3),
           ("gf2147483647", 2147483647, 1, None, 7), ("gf3e5", 3, 5, None, None), ("gf2e16", 2, 16, None, None),
           ("goldilocks", H.GOLDILOCKS, 1, None, 7), ("gf251e3", 251, 3, None, None), ("gf2e32", 2, 32, None, None)]


def _pair(tag):
    name, p, m, irr, alpha = next(f for f in _FIELDS if f[0] == tag)
    if m == 1:
        GF = ga.GF(p)
        return GF, O.OracleField(p, 1, None, int(GF.primitive_element))
    GF = ga.GF(p, m) if irr is None else ga.GF(p, m, irreducible_poly=irr)
    return GF, O.OracleField(p, m, int(GF.irreducible_poly), int(GF.primitive_element), lookup=GF.order <= 2**16)


def _rand(rng, q, shape):
    if q > 2**63:
        return (rng.integers(0, 2**63, shape, dtype=np.uint64) * 2 + rng.integers(0, 2, shape, dtype=np.uint64)) % np.uint64(q)
    return rng.integers(0, q, shape, dtype=np.uint64)


@pytest.mark.parametrize("tag", [f[0] for f in _FIELDS])
def test_matmul_against_oracle(tag):
    """Ragged sizes around the 64 x 64 x 16 tile, batch broadcasting, vectors (tests/fields/test_linalg.py:181-268)."""
    GF, F = _pair(tag)
    rng = np.random.default_rng(17)
    for M, K, N in [(1, 1, 1), (3, 5, 2), (64, 16, 64), (65, 17, 63), (130, 100, 70), (7, 257, 9)]:
        A, B = _rand(rng, GF.order, (M, K)), _rand(rng, GF.order, (K, N))
        H.assert_equal_ints((GF(A) @ GF(B)).numpy(), F.matmul(A, B), f"{tag} {M}x{K}x{N}")
    # stacks and broadcasting: (3, 1, M, K) @ (2, K, N) -> (3, 2, M, N); (K,) @ (K, N); (M, K) @ (K,)
    A, B = _rand(rng, GF.order, (3, 1, 5, 7)), _rand(rng, GF.order, (2, 7, 4))
    C = (GF(A) @ GF(B)).numpy()
    assert C.shape == (3, 2, 5, 4)
    for i in range(3):
        for j in range(2):
            H.assert_equal_ints(C[i, j], F.matmul(A[i, 0], B[j]))
    v, w = _rand(rng, GF.order, 7), _rand(rng, GF.order, 5)
    H.assert_equal_ints((GF(v) @ GF(B[0])).numpy(), F.matmul(v.reshape(1, 7), B[0])[0])
    H.assert_equal_ints((GF(A[0, 0]) @ GF(v)).numpy(), F.matmul(A[0, 0], v.reshape(7, 1))[:, 0])
    assert int(GF(v) @ GF(v)) == int(F.matmul(v.reshape(1, 7), v.reshape(7, 1))[0, 0])
    H.assert_equal_ints(np.dot(GF(A[0, 0]), GF(B[0])).numpy(), F.matmul(A[0, 0], B[0]))
    assert int(np.vdot(GF(v), GF(v))) == int(F.matmul(v.reshape(1, 7), v.reshape(7, 1))[0, 0])
    assert int(np.inner(GF(v), GF(v))) == int(F.matmul(v.reshape(1, 7), v.reshape(7, 1))[0, 0])
    H.assert_equal_ints(np.outer(GF(v), GF(w)).numpy(), F.matmul(v.reshape(7, 1), w.reshape(1, 5)))
    with pytest.raises(ValueError):
        GF(A[0, 0]) @ GF(w)


@pytest.mark.parametrize("tag", [f[0] for f in _FIELDS])
def test_elimination_against_oracle(tag):
    """row_reduce / PLU / LU / det / inv / solve / rank / subspaces on random, rank-deficient and rectangular matrices."""
    GF, F = _pair(tag)
    rng = np.random.default_rng(23)
    q = GF.order
    mats = [_rand(rng, q, (n, n)) for n in (1, 2, 3, 4, 9, 33)] + [_rand(rng, q, s) for s in ((3, 7), (8, 5), (20, 31))]
    low = _rand(rng, q, (12, 4))
    mats.append(np.asarray(F.matmul(low, _rand(rng, q, (4, 12)))))          # rank <= 4
    z = _rand(rng, q, (6, 6)); z[:, 0] = 0; z[2, :] = 0; mats.append(z)      # zero column and row: pivot skips
    sw = _rand(rng, q, (5, 5)); sw[0, 0] = 0; sw[1, 1] = 0; mats.append(sw)  # forces row exchanges
    for A in mats:
        gA = GF(A)
        rre, rank = F.row_reduce(A)
        H.assert_equal_ints(gA.row_reduce().numpy(), rre, "row_reduce")
        assert np.linalg.matrix_rank(gA) == F.matrix_rank(A)
        p, l, u = gA.plu_decompose()
        P, Lm, U, _ = F.plu_decompose(A)
        H.assert_equal_ints(p.numpy(), P); H.assert_equal_ints(l.numpy(), Lm); H.assert_equal_ints(u.numpy(), U)
        H.assert_equal_ints((p @ l @ u).numpy(), A, "PLU product")
        for op in ("row_space", "column_space", "left_null_space", "null_space"):
            want = getattr(F, op)(A)
            got = getattr(gA, op)().numpy()
            assert got.shape == want.shape or (got.size == 0 and want.size == 0), op
            if want.size:
                H.assert_equal_ints(got, want, op)
        if A.shape[0] == A.shape[1]:
            assert int(np.linalg.det(gA)) == F.det(A)
            try:
                want = F.inv(A)
            except np.linalg.LinAlgError:
                with pytest.raises(np.linalg.LinAlgError):
                    np.linalg.inv(gA)
            else:
                H.assert_equal_ints(np.linalg.inv(gA).numpy(), want, "inv")
                b = _rand(rng, q, (A.shape[0], 3))
                H.assert_equal_ints(np.linalg.solve(gA, GF(b)).numpy(), F.solve(A, b), "solve")
                H.assert_equal_ints(np.linalg.solve(gA, GF(b[:, 0])).numpy(), F.solve(A, b[:, 0]), "solve 1-D")
            if A.shape[0] - 1 <= A.shape[1]:
                try:
                    Lw, Uw = F.lu_decompose(A)
                except ValueError:
                    with pytest.raises(ValueError):
                        gA.lu_decompose()
                else:
                    lg, ug = gA.lu_decompose()
                    H.assert_equal_ints(lg.numpy(), Lw); H.assert_equal_ints(ug.numpy(), Uw)
    H.assert_equal_ints(GF(mats[3]).row_reduce(eye="right").numpy(), F.row_reduce(mats[3][::-1, ::-1])[0][::-1, ::-1])
    H.assert_equal_ints(GF(mats[7]).row_reduce(ncols=3).numpy(), F.row_reduce(mats[7], ncols=3)[0])


def test_batched_extensions_and_sizes():
    """A stack of 4096 GF(2^8) 16 x 16 systems (inverse, determinant, RREF in one launch each) and one 512 x 512 inverse."""
    GF, F = _pair("gf256")
    rng = np.random.default_rng(29)
    S = rng.integers(0, 256, (4096, 16, 16), dtype=np.uint8)
    gS = GF(S)
    dets = ga.linalg.det_batched(gS).numpy()
    rre, ranks = ga.linalg.row_reduce_batched(gS)
    for i in range(0, 4096, 257):
        assert int(dets[i]) == F.det(S[i])
        H.assert_equal_ints(rre.numpy()[i], F.row_reduce(S[i])[0])
    full = ranks == 16
    assert np.array_equal(dets != 0, full)
    inv = ga.linalg.inv_batched(GF(S[full]))
    prod = (GF(S[full]) @ inv).numpy()
    assert np.array_equal(prod, np.broadcast_to(np.eye(16, dtype=np.uint8), prod.shape))
    with pytest.raises(np.linalg.LinAlgError):
        ga.linalg.inv_batched(GF(np.zeros((2, 3, 3), dtype=np.uint8)))
    P = ga.GF(65537)
    A = rng.integers(0, 65537, (512, 512), dtype=np.uint32)
    gA = P(A)
    Ai = np.linalg.inv(gA)
    assert np.array_equal((gA @ Ai).numpy(), np.eye(512, dtype=np.uint32))


@pytest.mark.parametrize("tag,n", [("gf256", 384), ("gf65537", 400), ("gf2147483647", 370), ("gf3e5", 365)])
def test_wide_elimination_path(tag, n):
    """Matrices large enough for the two-kernels-per-column path: RREF, rank and inverse against the oracle, including a
    rank-deficient matrix and a rectangular one with ncols < n."""
    GF, F = _pair(tag)
    rng = np.random.default_rng(31)
    q = GF.order
    A = _rand(rng, q, (n, n))
    gA = GF(A)
    Ai = np.linalg.inv(gA)
    H.assert_equal_ints(Ai.numpy(), F.inv(A), "inv")
    assert np.array_equal((gA @ Ai).numpy().astype(np.uint64), np.eye(n, dtype=np.uint64))
    low = np.asarray(F.matmul(_rand(rng, q, (n, 7)), _rand(rng, q, (7, n + 40))))
    low[5] = 0
    rre, rank = F.row_reduce(low)
    H.assert_equal_ints(GF(low).row_reduce().numpy(), rre, "rank-deficient RREF")
    assert np.linalg.matrix_rank(GF(low)) == rank == 7
    R = _rand(rng, q, (300, 500))
    H.assert_equal_ints(GF(R).row_reduce(ncols=123).numpy(), F.row_reduce(R, ncols=123)[0], "ncols")
    stack = GF(np.stack([A, A[::-1].copy()]))
    rs, ranks = ga.linalg.row_reduce_batched(stack)
    assert list(ranks) == [n, n] and np.array_equal(rs.numpy()[0].astype(np.uint64), np.eye(n, dtype=np.uint64))


@pytest.mark.parametrize("p", [2, 3, 31, 127, 251])
def test_prime_field_matmul_on_matrix_cores(p):
    """Products large enough for the int8 MFMA path (centred residues, exact int32 accumulation): ragged sizes around the
    128 x 128 x 64 tiles, every storage dtype, stacks with a broadcast operand -- against the oracle's scalar loop."""
    GF = ga.GF(p)
    F = O.OracleField(p, 1, None, int(GF.primitive_element))
    rng = np.random.default_rng(p)
    for M, K, N in [(300, 500, 260), (128, 64, 128), (129, 65, 127), (1, 4096, 600), (513, 1000, 5)]:
        A, B = rng.integers(0, p, (M, K)), rng.integers(0, p, (K, N))
        want = F.matmul(A, B)
        for dt in GF.dtypes[:1] + GF.dtypes[-1:]:
            got = (GF(A.astype(dt), dtype=dt) @ GF(B.astype(dt), dtype=dt)).numpy()
            H.assert_equal_ints(got, want, f"GF({p}) {M}x{K}x{N} {np.dtype(dt).name}")
    A3, B1 = rng.integers(0, p, (3, 200, 300)), rng.integers(0, p, (300, 150))
    C = (GF(A3) @ GF(B1)).numpy()
    for i in range(3):
        H.assert_equal_ints(C[i], F.matmul(A3[i], B1))
    # extremes: all entries p - 1 (largest centred magnitude) over a long K
    A, B = np.full((130, 20000), p - 1), np.full((20000, 140), p - 1)
    assert np.all((GF(A) @ GF(B)).numpy() == (20000 * (p - 1) * (p - 1)) % p)


@pytest.mark.parametrize("p", [65537, 7340033, 2147483647, 257])
def test_large_prime_matmul_on_matrix_cores(p):
    """Primes above 256 split into 7-bit limbs (3 limbs below 2^21, 5 below 2^35): NL^2 exact int8 GEMMs whose int32 sums
    are folded mod p.  Sizes above the 2^27-MAC threshold, ragged edges, all-(p-1) operands, both storage dtypes."""
    GF = ga.GF(p)
    F = O.OracleField(p, 1, None, int(GF.primitive_element))
    rng = np.random.default_rng(p % 1000)
    for M, K, N in [(512, 512, 512), (515, 700, 381)]:
        A, B = rng.integers(0, p, (M, K)), rng.integers(0, p, (K, N))
        want = F.matmul(A, B)
        for dt in (GF.dtypes[0], np.int64):
            got = (GF(A.astype(dt), dtype=dt) @ GF(B.astype(dt), dtype=dt)).numpy()
            H.assert_equal_ints(got, want, f"GF({p}) {M}x{K}x{N} {np.dtype(dt).name}")
    A, B = np.full((600, 900), p - 1, dtype=np.int64), np.full((900, 600), p - 1, dtype=np.int64)
    assert np.all((GF(A) @ GF(B)).numpy().astype(np.int64) == (900 * pow(p - 1, 2, p)) % p)


@pytest.mark.parametrize("m", [2, 3, 4, 7, 8, 12, 16, 17, 20, 32])
def test_binary_extension_field_matmul_on_matrix_cores(m):
    """r06: GF(2^m), m <= 32, products with M, N >= 128 (and at least 2^24 multiply-adds): three (four, five) levels of Karatsuba over the
    bit positions -- 27 (81, at most 243) planes parity(A & mask_t), as many exact int8 GEMMs with the epilogue of GF(2) on the batch dimension of one
    launch, and a fold that xors r_t(x) mod f where the product bit is set (run_mfma_bits) -- against the oracle's table loop.  Degrees that
    are not powers of two (zero masks dropped), every storage dtype, all-ones operands over a long K, a stack with a broadcast operand, the
    hand-over to the other kernels below 128 rows; a subprocess with the planes switched OFF (GFA_MFMA_BITS_MIN_LOG=62) agrees on the same
    inputs."""
    GF = ga.GF(2**m)
    q = 2**m
    F = O.OracleField(2, m, int(GF.irreducible_poly), int(GF.primitive_element), lookup=m <= 16)
    rng = np.random.default_rng(m)
    shapes = ([(512, 300, 512), (300, 77, 257), (128, 1100, 129)] if m <= 16 else [(256, 64, 256), (129, 300, 130)]) + ([(1024, 1024, 1024), (1030, 1100, 1000)] if m in (8, 16) else [])  # (no oracle tables above 2^16 elements: small shapes)
    for M, K, N in shapes:
        A, B = rng.integers(0, q, (M, K)), rng.integers(0, q, (K, N))
        want = F.matmul(A, B)
        for dt in GF.dtypes[:1] + GF.dtypes[-1:]:
            got = (GF(A.astype(dt), dtype=dt) @ GF(B.astype(dt), dtype=dt)).numpy()
            H.assert_equal_ints(got, want, f"GF(2^{m}) {M}x{K}x{N} {np.dtype(dt).name}")
    # every bit set over a long K: the largest counts
    K = 70001
    A, B = np.full((256, K), q - 1), np.full((K, 256), q - 1)
    sq = int(F.mul(np.array([q - 1], dtype=np.uint64), np.array([q - 1], dtype=np.uint64))[0])
    assert np.all((GF(A) @ GF(B)).numpy() == (sq if K & 1 else 0))
    A3, B1 = (rng.integers(0, q, (3, 300, 400)), rng.integers(0, q, (400, 260))) if m <= 16 else (rng.integers(0, q, (2, 128, 130)), rng.integers(0, q, (130, 128)))
    C = (GF(A3) @ GF(B1)).numpy()
    for i in range(len(A3)):
        H.assert_equal_ints(C[i], F.matmul(A3[i], B1), f"stack {i}")
    # below 128 rows / columns the other kernels run: same values
    A, B = rng.integers(0, q, (127, 300)), rng.integers(0, q, (300, 2000 if m <= 16 else 200))
    H.assert_equal_ints((GF(A) @ GF(B)).numpy(), F.matmul(A, B), "below the tile size")


@pytest.mark.parametrize("order", [3**2, 3**3, 3**5, 5**4, 7**3, 3**7, 251**2, 13**5, 3**10, 5**8, 3**16])
def test_odd_characteristic_extension_field_matmul_on_matrix_cores(order):
    """r06: GF(p^m), odd p <= 251, m <= 16, products with M, N >= 128: Karatsuba over the base-p digit positions -- plane t of an operand is
    (sum of the digits in the leaf's set) mod p, one exact int8 GEMM mod p per leaf (all on the batch dimension of one launch), digit k of the
    result (sum_t P_t R_t[k]) mod p with R_t the leaf's weight polynomial reduced mod the field's polynomial on the host
    (run_mfma_digits) -- against the oracle's loop, for every storage dtype, degrees that are and are not powers of two, ragged shapes,
    all-(q - 1) operands over a long K, a stack with a broadcast operand."""
    GF = ga.GF(order)
    F = O.OracleField(GF.characteristic, GF.degree, int(GF.irreducible_poly), int(GF.primitive_element), lookup=order <= 2**16)
    rng = np.random.default_rng(order % 9973)
    small = order > 2**16  # (no oracle tables there: its explicit arithmetic sets the pace, so a sample of rows x columns is compared)
    shapes = ([(256, 256, 256), (130, 1100, 129)] if small else [(256, 64, 256), (300, 500, 257), (128, 1100, 129)]) + ([(512, 512, 512)] if order in (3**5, 251**2) else [])
    for M, K, N in shapes:
        A, B = rng.integers(0, order, (M, K)), rng.integers(0, order, (K, N))
        rows = np.array([0, 1, M // 2, M - 1]) if small else np.arange(M)
        cols = np.array([0, 1, N // 3, N - 1]) if small else np.arange(N)
        want = F.matmul(A[rows], B[:, cols])
        for dt in GF.dtypes[:1] + GF.dtypes[-1:]:
            got = (GF(A.astype(dt), dtype=dt) @ GF(B.astype(dt), dtype=dt)).numpy()
            H.assert_equal_ints(got[np.ix_(rows, cols)], want, f"GF({order}) {M}x{K}x{N} {np.dtype(dt).name}")
    K = 2001 if small else 20001
    A, B = np.full((128, K), order - 1), np.full((K, 130), order - 1)
    sq = int(F.mul(np.array([order - 1], dtype=np.uint64), np.array([order - 1], dtype=np.uint64))[0])
    one = F.matmul(np.full((1, K), order - 1), np.full((K, 1), order - 1))[0, 0]
    assert np.all((GF(A) @ GF(B)).numpy() == one), sq
    A3, B1 = rng.integers(0, order, (2, 200, 600)), rng.integers(0, order, (600, 150))
    C = (GF(A3) @ GF(B1)).numpy()
    for i in range(2):
        H.assert_equal_ints(C[i][:3], F.matmul(A3[i][:3], B1), f"stack {i}")


def test_binary_extension_field_matmul_agrees_with_the_table_kernels():
    """The same products with the bit planes switched off (GFA_MFMA_BITS_MIN_LOG is read once per process, so a child process runs the LDS
    product table / shift-and-xor kernels): both against the oracle."""
    import subprocess, sys, os

    code = (
        "import numpy as np, galois_amd as ga\n"
        "from oracle import gf_oracle as O\n"
        "rng = np.random.default_rng(3)\n"
        "for m in (3, 8, 16):\n"
        "    GF = ga.GF(2**m); q = 2**m\n"
        "    F = O.OracleField(2, m, int(GF.irreducible_poly), int(GF.primitive_element), lookup=True)\n"
        "    for M, K, N in [(256, 64, 256), (257, 65, 300), (512, 300, 259)]:\n"
        "        A, B = rng.integers(0, q, (M, K)), rng.integers(0, q, (K, N))\n"
        "        assert np.array_equal((GF(A) @ GF(B)).numpy().astype(np.uint64), F.matmul(A, B)), (m, M, K, N)\n"
        "print('ok')\n"
    )
    env = dict(os.environ, GFA_MFMA_BITS_MIN_LOG="62", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("p", [2**64 - 2**32 + 1, 2**61 - 1])
def test_64_bit_prime_matmul_on_matrix_cores(p):
    """r06: primes above 2^35 on the 7-bit limb path (ten limbs, 100 exact int8 GEMMs, 19 diagonals folded in 128-bit arithmetic) from 2^33
    multiply-adds: 2048 x 2100 x 2048 against Python integers on a sample of rows; all-(p - 1) operands."""
    import torch

    GF = ga.GF(p)
    rng = np.random.default_rng(p % 1000)
    M, K, N = 2048, 2100, 2048
    def rnd(shape):
        v = (rng.integers(0, 2**63, shape, dtype=np.uint64) % np.uint64(p >> 1)) * np.uint64(2) + rng.integers(0, 2, shape, dtype=np.uint64)
        return v  # < p, every bit position exercised
    a, b = rnd((M, K)), rnd((K, N))
    a[0, 0], a[1, 1], b[0, 0] = p - 1, 0, p - 1
    wrap = lambda v: GF._wrap(torch.from_numpy(v.view(np.int64)).cuda(), np.object_)  # (dtype=object in the reference: stored as uint64 words)
    A, B = wrap(a), wrap(b)
    C = (A @ B).numpy()
    Bo = b.astype(object)
    for i in (0, 2047):
        want = (a[i].astype(object) @ Bo) % p
        assert [int(v) for v in C[i]] == [int(v) for v in want], (p, i)
    ones_a, ones_b = np.full((256, 9000), p - 1, dtype=np.uint64), np.full((9000, 4096), p - 1, dtype=np.uint64)
    got = (wrap(ones_a) @ wrap(ones_b)).numpy()
    assert all(int(v) == (9000 * pow(p - 1, 2, p)) % p for v in got[::37, ::41].ravel())


def test_exceptions():
    """tests/fields/test_linalg.py:15-36, 86-92, 123-132, 291-299, 321-329, 345-353, 394-420."""
    GF = ga.GF(2**8)
    rng = np.random.default_rng(1)
    a1, a3 = GF(rng.integers(0, 256, 5)), GF(rng.integers(0, 256, (2, 2, 2)))
    for fn in (lambda x: x.row_reduce(), lambda x: x.lu_decompose(), lambda x: x.plu_decompose()):
        with pytest.raises(ValueError):
            fn(a1)
        with pytest.raises(ValueError):
            fn(a3)
    for fn in (np.linalg.inv, np.linalg.det):
        with pytest.raises(np.linalg.LinAlgError):
            fn(a1)
        with pytest.raises(np.linalg.LinAlgError):
            fn(a3)
    with pytest.raises(np.linalg.LinAlgError):
        np.linalg.solve(GF(rng.integers(0, 256, (2, 3))), GF(rng.integers(0, 256, 3)))
    with pytest.raises(TypeError):
        np.dot(a1, ga.GF(31)(np.arange(5)))
    with pytest.raises(ValueError):
        np.inner(a1, GF(rng.integers(0, 256, 4)))
    H2 = ga.GF(2)([[1, 0, 1, 0, 1, 0, 1, 0], [0, 1, 1, 0, 0, 1, 1, 0], [0, 0, 0, 1, 1, 1, 1, 0], [1, 1, 1, 1, 1, 1, 1, 1]])
    assert np.array_equal(H2.row_reduce(eye="right").numpy(), [[0, 1, 1, 1, 1, 0, 0, 0], [1, 0, 1, 1, 0, 1, 0, 0],
                                                              [1, 1, 0, 1, 0, 0, 1, 0], [1, 1, 1, 0, 0, 0, 0, 1]])
