"""Parity of polynomial evaluation / arithmetic, discrete logarithms, square roots, Vandermonde matrices and the `out=`
keyword against the reference's Sage fixtures and the oracle.  Bit-exact."""
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


def _coeffs(p):
    return [int(v) for v in p.coeffs.numpy()]


@pytest.mark.parametrize("tag", H.SAGE_POLYS)
def test_sage_poly_fixtures(tag):
    """tests/polys/test_arithmetic.py + test_operations.py: evaluate (element-wise and matrix), +, -, *, scalar *, d/dx."""
    props, d = H.load_sage_polys(tag)
    GF = _field(props)
    Y = GF(d["evaluate_Y"])
    for i in range(int(d["evaluate_count"])):
        poly = ga.Poly(d[f"evaluate{i}_X"], field=GF)
        z = poly(Y)
        assert type(z) is GF
        H.assert_equal_ints(z.numpy(), d[f"evaluate{i}_Z"], "evaluate")
        assert int(poly(Y[-1])) == int(d[f"evaluate{i}_Z"][-1])
    for X, Ym, Z in H.linalg_cases(d, "evaluate_matrix", "XYZ"):
        H.assert_equal_ints(ga.Poly(X, field=GF)(GF(Ym), elementwise=False).numpy(), Z, "evaluate_matrix")
    for op, fn in (("add", lambda a, b: a + b), ("subtract", lambda a, b: a - b), ("multiply", lambda a, b: a * b)):
        for X, Yp, Z in H.linalg_cases(d, op, "XYZ"):
            assert _coeffs(fn(ga.Poly(X, field=GF), ga.Poly(Yp, field=GF))) == H.as_int_list(Z), op
    for X, k, Z in H.linalg_cases(d, "scalar_multiply", "XYZ"):
        assert _coeffs(ga.Poly(X, field=GF) * int(k)) == H.as_int_list(Z), "scalar_multiply"
        assert _coeffs(int(k) * ga.Poly(X, field=GF)) == H.as_int_list(Z)
    for X, k, Z in H.linalg_cases(d, "derivative", "XYZ"):
        assert _coeffs(ga.Poly(X, field=GF).derivative(int(k))) == H.as_int_list(Z), "derivative"


@pytest.mark.parametrize("tag", H.SAGE_POLYS)
def test_sage_log_fixtures(tag):
    """tests/fields/test_advanced_arithmetic.py log vectors; fields without LOG tables raise NotImplementedError."""
    props, d = H.load_sage_polys(tag)
    GF = _field(props)
    x = GF(d["log_X"])
    z = np.log(x)  # LOG table for orders <= 2^20, Pohlig-Hellman on the device above
    H.assert_equal_ints(z, d["log_Z"], "log")
    assert np.array_equal(x.log(), z)
    assert np.array_equal((GF(GF.primitive_element) ** z).numpy(), x.numpy())
    with pytest.raises(ArithmeticError):
        np.log(GF([1, 0]))
    if GF.order > 4:
        # another primitive element as base (FieldArray.log docs, _fields/_array.py:2163-2179): beta = alpha^k, gcd(k, q-1) = 1
        k = next(k for k in range(2, GF.order) if np.gcd(k, GF.order - 1) == 1)
        beta = GF(GF.primitive_element) ** k
        zb = x.log(beta)
        assert np.array_equal((beta ** np.asarray(zb, dtype=np.int64)).numpy(), x.numpy())
        assert 0 <= zb.min() and zb.max() < GF.order - 1
    if (GF.order - 1) % 2 == 0 and GF.order > 3:
        with pytest.raises(ArithmeticError):
            x.log(GF(GF.primitive_element) ** 2)  # a square is not primitive


@pytest.mark.parametrize("q", [2, 7, 13, 17, 2**3, 3**3, 5**3, 2**8, 31, 65537, 3**5, 7340033, 2147483647, 2**16, 109**2])
def test_sqrt(q):
    """tests/fields/test_sqrt.py: y*y == x and y is the smaller of the two roots; non-squares raise ArithmeticError.
    The literal vectors of the reference's tests for GF(7), GF(13), GF(17) are included."""
    GF = ga.GF(q)
    rng = np.random.default_rng(q)
    x = GF.Random(2000, seed=int(q)) if q > 2000 else GF(np.arange(q))
    sq = x.is_square()
    xs = GF(x.numpy()[sq])
    y = np.sqrt(xs)
    assert type(y) is GF
    assert np.array_equal((y * y).numpy(), xs.numpy())
    yn, ynn = y.numpy().astype(np.uint64), (-y).numpy().astype(np.uint64)
    assert np.all(yn <= ynn)
    if q % 2 and q > 2:
        assert 0 < sq.sum() < sq.size or q <= 3
        with pytest.raises(ArithmeticError):
            np.sqrt(GF(x.numpy()[~sq][:3]))
    else:
        assert sq.all()
    if q == 7:
        assert np.array_equal(np.sqrt(GF([0, 1, 2, 4])).numpy(), [0, 1, 3, 2])
    if q == 13:
        assert np.array_equal(np.sqrt(GF([0, 1, 3, 4, 9, 10, 12])).numpy(), [0, 1, 4, 2, 3, 6, 5])
    if q == 17:
        assert np.array_equal(np.sqrt(GF([0, 1, 2, 4, 8, 9, 13, 15, 16])).numpy(), [0, 1, 6, 2, 5, 3, 8, 7, 4])


def test_vandermonde_power_outer_and_out():
    GF = ga.GF(2**8)
    F = O.OracleField(2, 8, 285, 2, lookup=True)
    V = GF.Vandermonde(GF.primitive_element, 7, 9)
    want = np.array([[int(F.pow([int(F.pow([2], [i])[0])], [j])[0]) for j in range(9)] for i in range(7)])
    assert V.shape == (7, 9) and np.array_equal(V.numpy(), want)
    x = GF([3, 7, 200])
    po = np.power.outer(x, np.array([[0, 1], [2, -1]]))
    assert po.shape == (3, 2, 2)
    for i, xv in enumerate([3, 7, 200]):
        assert np.array_equal(po.numpy()[i].ravel(), F.pow([xv] * 4, [0, 1, 2, -1]))
    with pytest.raises(ValueError):
        GF.Vandermonde(2, 0, 3)
    # out= (the result lands in the caller's array, _ufunc.py:309-319)
    a, b = GF.Random(1000, seed=1), GF.Random(1000, seed=2)
    out = GF.Zeros(1000)
    r = np.multiply(a, b, out=out)
    assert r is out and np.array_equal(out.numpy(), F.ufunc_u8(O.MUL, a.numpy(), b.numpy()))
    np.add(a, b, out=(out,))
    assert np.array_equal(out.numpy(), a.numpy() ^ b.numpy())
    with pytest.raises(ValueError):
        np.multiply(a, b, out=GF.Zeros(999))
    with pytest.raises(TypeError):
        np.multiply(a, b, out=np.zeros(1000, dtype=np.uint8))


def test_poly_evaluate_large_against_oracle():
    """Degree-255 polynomial at 2^20 points over GF(2^8) and GF(65537): Horner kernel vs the oracle's Horner."""
    rng = np.random.default_rng(41)
    for q, F in ((2**8, O.OracleField(2, 8, 285, 2, lookup=True)), (65537, O.OracleField(65537, 1, None, 3))):
        GF = ga.GF(q)
        c = rng.integers(1, q, 256)
        x = rng.integers(0, q, 1 << 20)
        got = ga.Poly(c, field=GF)(GF(x)).numpy()
        idx = rng.integers(0, 1 << 20, 4096)
        H.assert_equal_ints(got[idx], F.poly_eval(c, x[idx]))


@pytest.mark.parametrize("q", [2**8, 2**4, 2, 3**5, 31, 251, 7**2, 3])
def test_poly_evaluate_of_byte_fields_through_the_product_table_in_lds(q):
    """r06: fields of at most 256 elements, uint8 arrays of at least 65536 points, at least four coefficients: Horner's rule with the full
    64 KiB product table in LDS (row = the point), xor or the 64 KiB sum table (row = the coefficient) for the addition, four chains per lane
    (poly_eval_tab8_kernel) -- every point of every field value (all q values of x occur), lengths that leave the four chains ragged, zero
    and q - 1 coefficients, against the oracle's Horner; a short call (generic kernel) agrees."""
    GF = ga.GF(q)
    F = O.OracleField(GF.characteristic, GF.degree, int(GF.irreducible_poly) if GF.degree > 1 else None, int(GF.primitive_element), lookup=True)
    rng = np.random.default_rng(q)
    for ncoef, n in ((101, 65536 + 4099), (4, 300_001), (256, 1 << 18)):
        c = rng.integers(0, q, ncoef)
        c[0] = q - 1
        c[ncoef // 2] = 0
        x = rng.integers(0, q, n)
        x[:q] = np.arange(q)
        got = ga.Poly(c, field=GF)(GF(x.astype(np.uint8), dtype=np.uint8)).numpy()
        idx = np.concatenate([np.arange(q), np.arange(n - 4100, n), rng.integers(0, n, 3000)])
        H.assert_equal_ints(got[idx], F.poly_eval(c, x[idx]), f"GF({q}) {ncoef} coefficients at {n} points")
        short = ga.Poly(c, field=GF)(GF(x[:1000].astype(np.uint8), dtype=np.uint8)).numpy()
        H.assert_equal_ints(short, got[:1000], "generic kernel on a short array")


@pytest.mark.parametrize("q", [2**64 - 2**32 + 1, 2**61 - 1, 4294967291, 2**32, 2**40, 7340033, 3**16, 251**3, 2**63])
def test_discrete_log_without_tables(q):
    """Fields beyond the table limit: alpha ** log(x) == x, 0 <= log < q - 1, log(alpha^k) == k, other bases, log(0) raises."""
    GF = ga.GF(q)
    x = GF.Random(300, low=1, seed=3)
    z = np.log(x)
    zi = [int(v) for v in np.asarray(z).ravel()]
    assert all(0 <= v < q - 1 for v in zi)
    alpha = int(GF.primitive_element)
    F = O.OracleField(GF.characteristic, GF.degree, int(GF.irreducible_poly) if GF.degree > 1 else None, alpha)
    xs = x.numpy()
    for v, e in list(zip(xs.ravel(), zi))[:40]:
        # alpha ** e by square-and-multiply in the oracle (exponents up to 2^64 do not fit the int64 power kernel)
        r, b, ee = 1, alpha, e
        while ee:
            if ee & 1:
                r = int(F.mul([r], [b])[0])
            b = int(F.mul([b], [b])[0])
            ee >>= 1
        assert r == int(v)
    ks = [0, 1, 2, 12345, (q - 2) % (2**62)]
    pw = GF(alpha) ** np.array(ks, dtype=np.int64)
    assert [int(v) for v in np.asarray(np.log(pw)).ravel()] == [k % (q - 1) for k in ks]
    with pytest.raises(ArithmeticError):
        np.log(GF([1, 0]) if q < 2**63 else GF([1, 0]))
    import math

    k = next(k for k in range(3, 1000) if math.gcd(k, q - 1) == 1)
    beta = GF(alpha) ** k
    zb = [int(v) for v in np.asarray(x[:20].log(beta)).ravel()]
    for v, e in zip(xs[:20], zb):
        r, b, ee = 1, int(beta), e
        while ee:
            if ee & 1:
                r = int(F.mul([r], [b])[0])
            b = int(F.mul([b], [b])[0])
            ee >>= 1
        assert r == int(v)


@pytest.mark.parametrize("tag", H.SAGE_POLYS)
def test_sage_orders_trace_norm(tag):
    """tests/fields/test_arithmetic_methods.py: additive / multiplicative order, field trace and norm vs the Sage vectors."""
    props, d = H.load_sage_polys(tag)
    GF = _field(props)
    H.assert_equal_ints(GF(d["additive_order_X"]).additive_order(), d["additive_order_Z"])
    H.assert_equal_ints(GF(d["multiplicative_order_X"]).multiplicative_order(), d["multiplicative_order_Z"])
    tr = GF(d["field_trace_X"]).field_trace()
    assert type(tr) is GF.prime_subfield
    H.assert_equal_ints(tr.numpy(), d["field_trace_Z"])
    nm = GF(d["field_norm_X"]).field_norm()
    assert type(nm) is GF.prime_subfield
    H.assert_equal_ints(nm.numpy(), d["field_norm_Z"])
    with pytest.raises(ArithmeticError):
        GF([1, 0]).multiplicative_order()


def test_berlekamp_massey():
    """tests/test_berlekamp_massey.py: the Sage known answers (GF(2), GF(3), GF(2^3), GF(3^3) primitive LFSRs and a random
    GF(2) sequence), exceptions, and random / LFSR-generated sequences over larger fields against the oracle."""
    y = [0, 0, 1, 0, 0, 1, 1, 0, 1, 0, 1, 1, 1, 1, 0, 0, 0, 1, 0, 0, 1, 1, 0, 1, 0, 1, 1, 1, 1, 0, 0, 0, 1, 0, 0, 1, 1, 0, 1, 0, 1, 1, 1, 1, 0, 0, 0, 1, 0, 0]
    assert _coeffs(ga.berlekamp_massey(ga.GF(2)(y))) == [1, 0, 0, 1, 1]  # x^4 + x + 1
    y = [1, 1, 1, 1, 0, 0, 0, 1, 0, 0, 2, 1, 0, 1, 1, 1, 2, 0, 0, 2, 2, 0, 1, 0, 2, 2, 1, 1, 0, 1, 0, 1, 2, 1, 2, 2, 1, 2, 0, 1, 2, 2, 2, 2, 0, 0, 0, 2, 0, 0]
    assert _coeffs(ga.berlekamp_massey(ga.GF(3)(y))) == [1, 0, 0, 1, 2]  # x^4 + x + 2
    y = [1, 1, 1, 1, 2, 2, 2, 1, 4, 4, 7, 7, 3, 0, 5, 1, 5, 5, 5, 6, 1, 1, 2, 0, 2, 1, 6, 2, 7, 5, 3, 1, 7, 7, 4, 4, 5, 6, 3, 2, 2, 2, 7, 4, 4, 1, 6, 3, 6, 5]
    assert _coeffs(ga.berlekamp_massey(ga.GF(2**3)(y))) == [1, 0, 0, 1, 3]  # x^4 + x + 3
    y = [1, 1, 1, 1, 19, 19, 19, 1, 25, 25, 16, 4, 24, 6, 6, 6, 26, 2, 2, 9, 4, 11, 1, 11, 13, 21, 9, 9, 12, 10, 3, 0, 6, 2, 4, 3, 6, 15, 18, 7, 20, 20, 20, 8, 17, 17, 2, 1, 13, 19]
    assert _coeffs(ga.berlekamp_massey(ga.GF(3**3)(y))) == [1, 0, 0, 1, 10]  # x^4 + x + 10
    y = [0, 1, 0, 0, 1, 1, 0, 0, 0, 0, 1, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1, 1, 1, 0, 1, 0, 1, 1, 1, 1, 0, 0, 1, 0, 0, 1, 1, 1, 0, 0, 1, 1, 0, 0, 0, 0, 1, 0, 1]
    want = [0] * 25
    for dgr in (24, 21, 19, 18, 17, 15, 14, 13, 12, 11, 9, 8, 7, 6, 5, 4, 2, 1, 0):
        want[24 - dgr] = 1
    assert _coeffs(ga.berlekamp_massey(ga.GF(2)(y))) == want
    with pytest.raises(TypeError):
        ga.berlekamp_massey(np.array(y))
    with pytest.raises(ValueError):
        ga.berlekamp_massey(ga.GF(2)([y, y]))
    with pytest.raises(ValueError):
        ga.berlekamp_massey(ga.GF(2)(y), output="invalid-argument")
    rng = np.random.default_rng(5)
    for q in (2**8, 65537, 3**5, 2**64 - 2**32 + 1, 2**32):
        GF = ga.GF(q)
        F = O.OracleField(GF.characteristic, GF.degree, int(GF.irreducible_poly) if GF.degree > 1 else None, int(GF.primitive_element))
        for n in (1, 7, 64, 300):
            s = np.array([int(rng.integers(0, 2**62)) % q for _ in range(n)], dtype=np.uint64)
            if n == 64:  # a genuine LFSR sequence of order 9
                taps = [int(rng.integers(1, 2**62)) % q for _ in range(9)]
                seq = [int(v) for v in s[:9]]
                for _ in range(n - 9):
                    acc = 0
                    for tp, v in zip(taps, seq[-9:]):
                        acc = int(F.add([acc], [int(F.mul([tp], [v])[0])])[0])
                    seq.append(acc)
                s = np.array(seq, dtype=np.uint64)
            gs = GF([int(v) for v in s]) if q > 2**63 else GF(s)
            got = _coeffs(ga.berlekamp_massey(gs, output="connection"))
            assert got == [int(v) for v in F.berlekamp_massey(s)], (q, n)
            minimal = _coeffs(ga.berlekamp_massey(gs))
            assert minimal[0] == 1 and minimal == (got[::-1] if got[-1] == 1 else minimal)
