"""`where=` and `initial=` on ufuncs and their methods (VERDICT r04 missing #2).

The reference forwards every keyword to the NumPy ufunc it built for the field (_domains/_ufunc.py:349, 364, 379, 403, 418),
so the contract is NumPy's semantics on the integer values: masked-out positions of a call are not computed (they keep the
`out` array's value and raise nothing), masked-out elements of a reduction do not take part, the fold starts from `initial`.
Expected values: Python integers for prime fields, xor for GF(2^m) sums, and -- for fields whose arithmetic has no one-line
model -- the unmasked device result (pinned to the oracle by the other suites) blended / compressed by NumPy on the host.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import galois_amd as ga  # noqa: E402


def ints(a):
    return np.array([int(v) for v in np.asarray(a.numpy()).ravel()], dtype=object).reshape(a.shape)


FIELDS = [7, 2**8, 3**5, 65537, 2**32, 2**64 - 2**32 + 1, 2**100, 36893488147419103183]


@pytest.mark.parametrize("order", FIELDS)
def test_where_on_calls_blends_with_out_and_computes_nothing_else(order):
    GF = ga.GF(order)
    x = GF.Random((6, 50), seed=1)
    y = GF.Random((6, 50), low=1, seed=2)
    y[0, :5] = 0  # divisors that are zero ...
    mask = np.ones((6, 50), dtype=bool)
    mask[0, :5] = False  # ... only where nothing is computed
    mask[3, ::3] = False
    hx, hy = ints(x), ints(y)
    y_safe = y.copy()
    y_safe[0, :5] = 1
    for uf, full in ((np.add, x + y), (np.subtract, x - y), (np.multiply, x * y), (np.true_divide, x / y_safe)):
        old = GF.Random((6, 50), seed=3)
        want = np.where(mask, ints(full), ints(old))
        got = uf(x, y, where=mask, out=old)
        assert got is old and np.array_equal(ints(got), want), uf.__name__
        fresh = uf(x, y, where=mask)  # no `out`: NumPy leaves the holes uninitialised; here they are zero
        assert type(fresh) is GF and np.array_equal(ints(fresh)[mask], ints(full)[mask]) and not ints(fresh)[~mask].any()
    # the mask takes part in broadcasting
    row_mask = mask[3]
    got = np.add(x, y, where=row_mask)
    assert np.array_equal(ints(got)[:, row_mask], ints(x + y)[:, row_mask])
    got = np.multiply(x[0], y[1], where=mask)
    assert got.shape == (6, 50) and np.array_equal(ints(got)[mask], np.broadcast_to(ints(x[0] * y[1]), (6, 50))[mask])
    # unary, power, scalar multiplication
    z = y.copy()
    want = np.where(mask, ints(np.reciprocal(y_safe)), ints(z))
    assert np.array_equal(ints(np.reciprocal(y, where=mask, out=z)), want)
    assert np.array_equal(ints(np.negative(x, where=mask))[mask], ints(-x)[mask])
    assert np.array_equal(ints(np.power(y, -3, where=mask))[mask], ints(y_safe ** -3)[mask])  # 0 ** -3 only where masked out
    assert np.array_equal(ints(np.multiply(x, 5, where=mask))[mask], ints(x * 5)[mask])
    assert np.array_equal(ints(np.square(x, where=mask))[mask], ints(x * x)[mask])
    with pytest.raises(ZeroDivisionError):
        np.true_divide(x, y, where=np.ones((6, 50), dtype=bool))
    with pytest.raises(TypeError):
        np.add(x, y, where=np.ones((6, 50), dtype=np.int64))
    with pytest.raises(ValueError):
        np.add(x, y, where=mask, out=GF.Zeros((6, 49)))
    # ufunc.outer forwards the keyword to the call
    a, b = x[0, :4], y[1, :3]
    om = np.array([[True, False, True]] * 4)
    assert np.array_equal(ints(np.multiply.outer(a, b, where=om))[om], ints(np.multiply.outer(a, b))[om])
    # methods NumPy does not give the keywords to
    for bad in (lambda: np.add.accumulate(x, where=mask), lambda: np.add(x, y, initial=1), lambda: np.add.reduceat(x, [0, 2], initial=1)):
        with pytest.raises(TypeError):
            bad()


@pytest.mark.parametrize("p", [7, 65537, 4294967291, 2**61 - 1, 2**64 - 2**32 + 1, 36893488147419103183])
def test_reduce_with_where_and_initial_in_prime_fields_against_python_integers(p):
    GF = ga.GF(p)
    rng = np.random.default_rng(p % 1000)
    h = np.array([int(v) % p for v in rng.integers(0, 2**63, 60)], dtype=object).reshape(5, 12)
    h[h == 0] = 1
    x = GF(h)
    mask = rng.integers(0, 2, (5, 12)).astype(bool)
    mask[2] = False  # a row with nothing selected: the result is `initial`
    init = int(rng.integers(1, min(p, 2**62)))
    for axis in (0, 1, -1):
        ax = axis % 2
        sel = [[int(h[i, j]) for i in range(5) if mask[i, j]] for j in range(12)] if ax == 0 else [[int(v) for v, m in zip(h[i], mask[i]) if m] for i in range(5)]
        want_add = [(init + sum(s)) % p for s in sel]
        want_sub = [(init - sum(s)) % p for s in sel]
        want_mul, want_div = [], []
        for s in sel:
            prod = 1
            for v in s:
                prod = prod * v % p
            want_mul.append(init * prod % p)
            want_div.append(init * pow(prod, -1, p) % p)
        assert [int(v) for v in ints(np.add.reduce(x, axis=axis, where=mask, initial=init))] == want_add
        assert [int(v) for v in ints(np.subtract.reduce(x, axis=axis, where=mask, initial=init))] == want_sub
        assert [int(v) for v in ints(np.multiply.reduce(x, axis=axis, where=mask, initial=GF(init)))] == want_mul
        assert [int(v) for v in ints(np.true_divide.reduce(x, axis=axis, where=mask, initial=init))] == want_div
    # initial alone, keepdims, and np.sum / np.prod (which NumPy routes to add.reduce / multiply.reduce with the same keywords)
    assert [int(v) for v in ints(np.add.reduce(x, axis=1, initial=init))] == [(init + sum(int(v) for v in r)) % p for r in h]
    kd = np.add.reduce(x, axis=1, where=mask, initial=init, keepdims=True)
    assert kd.shape == (5, 1)
    assert int(np.sum(x, where=mask, initial=init)) == (init + sum(int(v) for v in h[mask])) % p
    tot = 1
    for v in h[mask]:
        tot = tot * int(v) % p
    assert int(np.prod(x, where=mask, initial=init)) == init * tot % p
    assert [int(v) for v in ints(np.sum(x, axis=0, where=mask, initial=init))] == [(init + sum(int(h[i, j]) for i in range(5) if mask[i, j])) % p for j in range(12)]
    # an empty axis: the result is the seed
    e = GF(np.zeros((3, 0), dtype=np.int64)) if p < 2**63 else GF(np.zeros((3, 0), dtype=object))
    assert [int(v) for v in ints(np.add.reduce(e, axis=1, initial=init))] == [init] * 3
    # the reference's ufuncs carry no identity: a mask without a seed is NumPy's ValueError
    with pytest.raises(ValueError, match="does not have an identity"):
        np.add.reduce(x, axis=0, where=mask)
    with pytest.raises(ValueError):
        np.add.reduce(x, axis=0, initial=p)  # the seed must be a field element
    with pytest.raises(ZeroDivisionError):
        z = x.copy()
        z[1, 1] = 0
        np.true_divide.reduce(z, axis=1, where=np.ones((5, 12), dtype=bool), initial=init)


@pytest.mark.parametrize("m", [1, 8, 16, 32, 100])
def test_reduce_with_where_in_binary_fields_is_xor_and_needs_no_seed(m):
    """Characteristic 2: the reference's add / subtract ARE np.bitwise_xor (_fields/_ufunc.py:59-61), which has an identity."""
    GF = ga.GF(2**m)
    rng = np.random.default_rng(m)
    h = np.array([int(v) % (2**m) for v in rng.integers(0, 2**63, 70)], dtype=object).reshape(7, 10)
    x = GF(h)
    mask = rng.integers(0, 2, (7, 10)).astype(bool)
    mask[:, 4] = False
    for uf in (np.add, np.subtract):
        want = []
        for j in range(10):
            acc = 0
            for i in range(7):
                if mask[i, j]:
                    acc ^= int(h[i, j])
            want.append(acc)
        assert [int(v) for v in ints(uf.reduce(x, axis=0, where=mask))] == want
        assert [int(v) for v in ints(uf.reduce(x, axis=0, where=mask, initial=1))] == [w ^ 1 for w in want]
    if m > 1:
        with pytest.raises(ValueError, match="does not have an identity"):
            np.multiply.reduce(x, axis=0, where=mask)
    # products: the masked fold equals the fold over the compressed column, seeded
    col = 2
    seed = GF(3 % (2**m) or 1)
    got = np.multiply.reduce(x, axis=0, where=mask, initial=seed)[col]
    sel = GF(h[:, col][mask[:, col]]) if mask[:, col].any() else None
    want = seed * np.multiply.reduce(sel) if sel is not None else seed
    assert int(got) == int(want)


def test_masked_composites_and_host_results():
    """np.sqrt (several kernels) and np.log (an integer ndarray comes back) take the mask as well."""
    GF = ga.GF(7340033)
    rng = np.random.default_rng(3)
    r = GF(rng.integers(1, 7340033, 400, dtype=np.uint32))
    sq = r * r
    mask = rng.integers(0, 2, 400).astype(bool)
    nonres = GF(np.full(400, GF._primitive_element_int, dtype=np.uint32))  # a non-residue: np.sqrt raises on it
    x = GF(np.where(mask, sq.numpy(), nonres.numpy()).astype(np.uint32))
    got = np.sqrt(x, where=mask)
    assert np.array_equal(got.numpy()[mask], np.sqrt(sq).numpy()[mask])
    G8 = ga.GF(2**8)
    v = G8(rng.integers(0, 256, 300, dtype=np.uint8))
    m8 = v.numpy() != 0  # log 0 raises: mask it out
    lg = np.log(v, where=m8)
    assert isinstance(lg, np.ndarray) and np.array_equal(lg[m8], np.log(G8(v.numpy()[m8])))
    dst = np.full(300, -1, dtype=np.int64)
    np.log(v, where=m8, out=dst)
    assert np.array_equal(dst[m8], lg[m8]) and (dst[~m8] == -1).all()


@pytest.mark.parametrize("order,dt,wide", [(65521, np.uint16, np.int64), (65521, np.uint16, np.uint32), (2**32, np.uint32, np.int64),
                                             (3**10, np.uint16, np.uint32)])
def test_where_with_out_of_a_wider_dtype_keeps_values_above_the_sign_bit(order, dt, wide):
    """uint16 / uint32 field arrays are int16 / int32 bit patterns on the device, so blending a masked result into an
    `out` array of a wider dtype must not sign-extend -- every element >= 2^15 (2^31) used to come out negative."""
    GF = ga.GF(order)
    rng = np.random.default_rng(11)
    hi = order - 1 - rng.integers(0, min(1000, order // 4), 300)  # operands near the order: above the storage's sign bit
    x = GF(hi.astype(dt), dtype=dt)
    zero = GF(np.zeros(300, dtype=dt), dtype=dt)
    mask = rng.integers(0, 2, 300).astype(bool)
    out = GF(np.full(300, 5, dtype=wide), dtype=wide)
    got = np.add(x, zero, where=mask, out=out)
    assert got is out and out.dtype == wide
    want = np.where(mask, hi, 5)
    assert np.array_equal(ints(out), np.array([int(v) for v in want], dtype=object))
    s = np.add.reduce(x, where=mask, initial=GF(0)) if GF.characteristic != 2 else None
    if s is not None:
        assert int(s) == int(sum(int(v) for v in hi[mask]) % order) or GF.degree > 1


@pytest.mark.parametrize("order", [2**8, 65537, 2**100])
def test_out_keyword_of_array_functions(order):
    """`out=` on the linear-algebra and stacking array functions: the reference copies the result into the caller's array
    (_domains/_linalg.py:269-274); same here, for any storage width of the field, with NumPy's shape check."""
    GF = ga.GF(order)
    A, B = GF.Random((5, 7), seed=1), GF.Random((7, 4), seed=2)
    want = A @ B
    C = GF.Zeros((5, 4))
    got = np.dot(A, B, out=C)
    assert got is C and np.array_equal(ints(C), ints(want))
    D = GF.Zeros((5, 4))
    assert np.matmul(A, B, out=D) is D and np.array_equal(ints(D), ints(want))
    v, w = GF.Random(9, seed=3), GF.Random(9, seed=4)
    O2 = GF.Zeros((9, 9))
    assert np.outer(v, w, out=O2) is O2 and np.array_equal(ints(O2), ints(np.multiply.outer(v, w)))
    cat = GF.Zeros(18)
    assert np.concatenate([v, w], out=cat) is cat and np.array_equal(ints(cat), np.concatenate([ints(v), ints(w)]))
    st = GF.Zeros((2, 9))
    assert np.stack([v, w], out=st) is st and np.array_equal(ints(st), np.stack([ints(v), ints(w)]))
    with pytest.raises(ValueError):
        np.dot(A, B, out=GF.Zeros((4, 5)))
    with pytest.raises(TypeError):
        np.dot(A, B, out=np.zeros((5, 4), dtype=np.int64))
    if order == 65537:  # a wider storage dtype of the same field as the target
        W = GF(np.zeros((5, 4), dtype=np.int64), dtype=np.int64)
        assert np.dot(A, B, out=W) is W and W.dtype == np.int64 and np.array_equal(ints(W), ints(want))
