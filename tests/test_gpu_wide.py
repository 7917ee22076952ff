"""Fields of order 2^64 <= q < 2^128 on the device (galois_amd/_wide.py, csrc/gfa_wide.hip): the reference's three big Sage
folders table for table (tests/fields/test_arithmetic.py:17-199 runs them as dtype=object arrays), random and edge-case
operands against the Python-integer oracle, and the array surface (construction, indexing, broadcasting, errors)."""
import json
import os
import random

import numpy as np
import pytest

import galois_amd as ga
from oracle.wide_oracle import WideOracle
from tests import helpers as H

pytestmark = pytest.mark.gpu

TAGS = ["GF_2e100", "GF_36893488147419103183", "GF_109987e4"]


def _load(tag):
    d = np.load(os.path.join(H.GOLDEN, f"sage_wide_{tag}.npz"))
    props = json.loads(str(d["properties"]))
    p, m = props["characteristic"], props["degree"]
    if m == 1:
        GF = ga.GF(p, primitive_element=props["primitive_element"])
    else:
        GF = ga.GF(p, m, irreducible_poly=props["irreducible_poly"], primitive_element=props["primitive_element"])
    return GF, WideOracle(p, m, props["irreducible_poly"] if m > 1 else None), d, props


def _obj(a):
    a = np.asarray(a)
    return np.array([int(v) for v in a.ravel()], dtype=object).reshape(a.shape)


@pytest.mark.parametrize("tag", TAGS)
def test_sage_vectors_of_the_big_fields(tag):
    GF, W, d, props = _load(tag)
    assert GF.order == props["order"] and GF.dtypes == [np.object_] and GF.characteristic == props["characteristic"]
    for op, fn in [("add", np.add), ("subtract", np.subtract), ("multiply", np.multiply), ("divide", np.true_divide)]:
        x, y = GF(_obj(d[f"{op}_X"])), GF(_obj(d[f"{op}_Y"]))
        z = fn(x.reshape(-1, 1), y.reshape(1, -1))
        assert type(z) is GF and z.dtype == np.dtype(object) and z.shape == (x.size, y.size)
        H.assert_equal_ints(z.numpy(), _obj(d[f"{op}_Z"]), f"{tag} {op}")
    x = GF(_obj(d["additive_inverse_X"]))
    H.assert_equal_ints((-x).numpy(), _obj(d["additive_inverse_Z"]))
    x = GF(_obj(d["multiplicative_inverse_X"]))
    H.assert_equal_ints(np.reciprocal(x).numpy(), _obj(d["multiplicative_inverse_Z"]))
    H.assert_equal_ints((x ** -1).numpy(), _obj(d["multiplicative_inverse_Z"]))
    x = GF(_obj(d["power_X"]))
    z = x.reshape(-1, 1) ** _obj(d["power_Y"]).reshape(1, -1)  # exponents of +-100 bits
    H.assert_equal_ints(z.numpy(), _obj(d["power_Z"]), f"{tag} power")
    x = GF(_obj(d["scalar_multiply_X"]))
    z = x.reshape(-1, 1) * _obj(d["scalar_multiply_Y"]).reshape(1, -1)
    H.assert_equal_ints(z.numpy(), _obj(d["scalar_multiply_Z"]), f"{tag} scalar multiply")


@pytest.mark.parametrize("tag", TAGS)
def test_random_and_edge_operands_against_the_oracle(tag):
    GF, W, d, props = _load(tag)
    q, p = GF.order, GF.characteristic
    rng = random.Random(7)
    edge = [0, 1, 2, q - 1, q - 2, p - 1 if p < q else q - 1, (1 << 64) - 1, 1 << 64, (1 << 64) + 1, q // 2, q // 3]
    edge = [v % q for v in edge]
    a = edge + [rng.randrange(q) for _ in range(400)]
    b = [rng.choice(edge) for _ in range(len(edge))] + [rng.randrange(q) for _ in range(400)]
    A, B = GF(np.array(a, dtype=object)), GF(np.array(b, dtype=object))
    assert [int(v) for v in (A + B).numpy()] == [W.add(x, y) for x, y in zip(a, b)]
    assert [int(v) for v in (A - B).numpy()] == [W.sub(x, y) for x, y in zip(a, b)]
    assert [int(v) for v in (A * B).numpy()] == [W.mul(x, y) for x, y in zip(a, b)]
    assert [int(v) for v in (-A).numpy()] == [W.neg(x) for x in a]
    bnz = [y if y else 1 for y in b]
    Bnz = GF(np.array(bnz, dtype=object))
    assert [int(v) for v in (A / Bnz).numpy()] == [W.div(x, y) for x, y in zip(a, bnz)]
    assert [int(v) for v in np.reciprocal(Bnz).numpy()] == [W.inv(y) for y in bnz]
    es = [0, 1, 2, -1, -2, q - 1, q, -(q - 1), 3 * q + 5] + [rng.randrange(-q, q) for _ in range(40)]
    sub = a[:len(es)]
    subnz = [x if x else 3 for x in sub]
    got = GF(np.array(subnz, dtype=object)) ** np.array(es, dtype=object)
    assert [int(v) for v in got.numpy()] == [W.pow(x, e) for x, e in zip(subnz, es)]
    assert int(GF(0) ** 0) == 1 and int(GF(0) ** 5) == 0 and int(np.square(GF(q - 1))) == W.mul(q - 1, q - 1)
    assert [int(v) for v in (A * 12345678901234567890123).numpy()] == [W.mul(x, 12345678901234567890123 % p) for x in a]
    # broadcasting against a scalar, and the reference's error behaviour
    assert [int(v) for v in (A * GF(7)).numpy()] == [W.mul(x, 7 % q) for x in a]
    with pytest.raises(ZeroDivisionError):
        A / GF(np.array([0] * len(a), dtype=object))
    with pytest.raises(ZeroDivisionError):
        np.reciprocal(GF(0))
    with pytest.raises(ZeroDivisionError):
        GF(0) ** -3
    with pytest.raises(ValueError):
        GF(q)
    with pytest.raises(ValueError):
        GF(-1)
    with pytest.raises(TypeError):
        A + 1
    with pytest.raises(TypeError):
        GF([1, 2], dtype=np.int64)
    # (reduceat / at / log / sqrt / fft on these fields: test_fft_sqrt_reduceat_and_at_on_the_big_fields_against_the_oracle)
    assert int(np.add.reduceat(A, [0, len(a) - 1])[1]) == a[-1]


def test_array_surface_of_a_big_field():
    GF, W, d, props = _load("GF_36893488147419103183")
    q = GF.order
    x = GF(np.array([[1, 2, q - 1], [q - 2, 5, 0]], dtype=object))
    assert x.shape == (2, 3) and x.ndim == 2 and x.size == 6 and len(x) == 2
    assert int(x[0, 2]) == q - 1 and [int(v) for v in x[1].numpy()] == [q - 2, 5, 0]
    assert x.T.shape == (3, 2) and int(x.T[2, 0]) == q - 1
    y = x.reshape(3, 2)
    assert y.shape == (3, 2) and [int(v) for v in y.flatten().numpy()] == [1, 2, q - 1, q - 2, 5, 0]
    x[0, 0] = q - 5
    assert int(x[0, 0]) == q - 5
    assert (x == x.copy()).all() and not (x == GF.Zeros((2, 3))).all()
    z = np.add.outer(GF(np.array([1, 2], dtype=object)), GF(np.array([q - 1, 0, 3], dtype=object)))
    assert z.shape == (2, 3) and [int(v) for v in z.numpy().ravel()] == [0, 1, 4, 1, 2, 5]
    qd, r = divmod(x, GF(np.array(3, dtype=object)))
    assert [int(v) for v in r.numpy().ravel()] == [0] * 6 and [int(v) for v in qd.numpy().ravel()] == [W.div(int(v), 3) for v in x.numpy().ravel()]
    out = GF.Zeros((2, 3))
    r2 = np.multiply(x, x, out=out)
    assert r2 is out and [int(v) for v in out.numpy().ravel()] == [W.mul(int(v), int(v)) for v in x.numpy().ravel()]
    rr = GF.Random((4, 5), seed=3)
    assert rr.shape == (4, 5) and all(0 <= int(v) < q for v in rr.numpy().ravel())
    assert repr(GF(np.array([1, 2], dtype=object))).startswith("GF([1, 2]")
    with pytest.raises(NotImplementedError):  # (orders above 2^128: tests/test_gpu_big.py; extension fields there keep p < 2^32)
        ga.GF(2**127 - 1, 2, irreducible_poly=[1, 0, 1], primitive_element=[1, 3], verify=False)


def test_data_movement_functions_never_touch_the_limb_axis():
    """FieldArray.__array_function__ moves tensors with one entry per element; for order >= 2^64 the storage has a trailing limb
    axis, which np.flip / transpose / roll / repeat / take / concatenate(axis=-1) ... must not see (flip without an
    axis swapped the two limbs of every element)."""
    GF = ga.GF(2**100)
    q = GF.order
    rng = np.random.default_rng(1)
    h = np.array([int(rng.integers(0, 2**62)) * int(rng.integers(0, 2**38)) % q for _ in range(24)] , dtype=object).reshape(2, 3, 4)
    h[0, 0, 0], h[1, 2, 3] = q - 1, 1 << 64
    x = GF(h)
    eq = lambda a, b: a.shape == b.shape and all(int(u) == int(v) for u, v in zip(np.asarray(a.numpy()).ravel(), b.ravel()))
    assert eq(np.flip(x), np.flip(h)) and eq(np.flip(x, axis=-1), np.flip(h, axis=-1)) and eq(np.flip(x, axis=(0, 2)), np.flip(h, axis=(0, 2)))
    assert eq(np.transpose(x), np.transpose(h)) and eq(np.transpose(x, (1, 0, 2)), np.transpose(h, (1, 0, 2)))
    assert eq(np.swapaxes(x, 0, -1), np.swapaxes(h, 0, -1)) and eq(np.moveaxis(x, -1, 0), np.moveaxis(h, -1, 0))
    assert eq(np.roll(x, 5), np.roll(h, 5)) and eq(np.roll(x, 1, axis=-1), np.roll(h, 1, axis=-1))
    assert eq(np.repeat(x, 2), np.repeat(h, 2)) and eq(np.repeat(x, 3, axis=-1), np.repeat(h, 3, axis=-1))
    assert eq(np.take(x, [3, 0, 23]), np.take(h, [3, 0, 23])) and eq(np.take(x, [3, 0], axis=-1), np.take(h, [3, 0], axis=-1))
    assert eq(np.concatenate([x, x], axis=-1), np.concatenate([h, h], axis=-1)) and eq(np.concatenate([x, x]), np.concatenate([h, h]))
    assert eq(np.concatenate([x, x], axis=None), np.concatenate([h, h], axis=None))
    assert eq(np.stack([x, x], axis=-1), np.stack([h, h], axis=-1)) and eq(np.expand_dims(x, -1), np.expand_dims(h, -1))
    assert eq(np.tile(x[0, 0], 3), np.tile(h[0, 0], 3)) and eq(np.diag(x[0, 0]), np.diag(h[0, 0]) if False else np.array(
        [[h[0, 0][i] if i == j else 0 for j in range(4)] for i in range(4)], dtype=object))
    assert eq(np.diagonal(x[0]), np.diagonal(h[0])) and eq(np.tril(x[0]), np.tril(h[0])) and eq(np.triu(x[0], 1), np.triu(h[0], 1))
    cond = np.arange(24).reshape(2, 3, 4) % 3 == 0
    assert eq(np.where(cond, x, GF.Zeros((2, 3, 4))), np.where(cond, h, 0))
    assert eq(np.squeeze(x[:1]), h[0]) and eq(np.broadcast_to(x[0, 0], (5, 4)), np.broadcast_to(h[0, 0], (5, 4)))
    assert eq(np.append(x[0, 0], x[1, 1]), np.append(h[0, 0], h[1, 1])) and eq(np.delete(x[0, 0], 1), np.delete(h[0, 0], 1))
    for v in np.asarray(np.flip(x).numpy()).ravel():
        assert 0 <= int(v) < q
    with pytest.raises(NotImplementedError):
        np.sort(x)


@pytest.mark.parametrize("tag", TAGS)
def test_sage_convolve_and_matrix_multiply_of_the_big_fields(tag):
    """convolve.pkl and matrix_multiply.pkl of the three Sage folders of order >= 2^64 (the reference runs them as object-dtype
    loops: _domains/_function.py:141-167, _domains/_linalg.py:286-308): 16 / 16 field folders for rows f1 and f2."""
    GF, W, d, props = _load(tag)
    for i in range(int(d["convolve_count"])):
        x, y = GF(_obj(d[f"convolve_X_{i}"])), GF(_obj(d[f"convolve_Y_{i}"]))
        z = np.convolve(x, y)
        assert type(z) is GF
        H.assert_equal_ints(z.numpy(), _obj(d[f"convolve_Z_{i}"]), f"{tag} convolve {i}")
    for i in range(int(d["matrix_multiply_count"])):
        x, y = GF(_obj(d[f"matrix_multiply_X_{i}"])), GF(_obj(d[f"matrix_multiply_Y_{i}"]))
        z = x @ y
        assert type(z) is GF and z.shape == (x.shape[0], y.shape[1])
        H.assert_equal_ints(z.numpy(), _obj(d[f"matrix_multiply_Z_{i}"]), f"{tag} matmul {i}")
        H.assert_equal_ints(np.matmul(x, y).numpy(), _obj(d[f"matrix_multiply_Z_{i}"]))


@pytest.mark.parametrize("tag", TAGS)
def test_reductions_of_the_big_fields_against_the_oracle(tag):
    """ufunc.reduce / accumulate (left folds for subtract and divide), np.sum / prod / cumsum, vector products and batched
    matrix products against the Python-integer oracle."""
    GF, W, d, props = _load(tag)
    q = GF.order
    rnd = random.Random(7)
    h = np.array([rnd.randrange(1, q) for _ in range(3 * 700)], dtype=object).reshape(3, 700)
    h[0, 0], h[1, 5], h[2, 699] = q - 1, 1, q - 2
    x = GF(h)
    fold = {np.add: W.add, np.subtract: W.sub, np.multiply: W.mul, np.true_divide: W.div}

    def left_fold(f, row):
        acc = int(row[0])
        out = [acc]
        for v in row[1:]:
            acc = f(acc, int(v))
            out.append(acc)
        return out

    for uf, f in fold.items():
        want = [left_fold(f, h[r]) for r in range(3)]
        H.assert_equal_ints(uf.reduce(x, axis=1).numpy(), np.array([w[-1] for w in want], dtype=object), f"{tag} {uf.__name__}.reduce")
        H.assert_equal_ints(uf.accumulate(x[:, :40], axis=1).numpy(), np.array([left_fold(f, h[r, :40]) for r in range(3)], dtype=object))
    col = [left_fold(W.add, h[:, c])[-1] for c in range(700)]
    H.assert_equal_ints(np.add.reduce(x, axis=0).numpy(), np.array(col, dtype=object))
    assert np.add.reduce(x, axis=1, keepdims=True).shape == (3, 1)
    assert int(np.sum(x)) == left_fold(W.add, h.ravel())[-1] and int(np.prod(x[0, :50])) == left_fold(W.mul, h[0, :50])[-1]
    H.assert_equal_ints(np.cumsum(x[1, :30]).numpy(), np.array(left_fold(W.add, h[1, :30]), dtype=object))
    assert int(np.add.reduce(x[0, :1])) == int(h[0, 0])
    # empty reduction axis: identities for add / multiply, ValueError for subtract / divide (NumPy's rule, the reference's too)
    e = x[:, :0]
    H.assert_equal_ints(np.add.reduce(e, axis=1).numpy(), np.array([0, 0, 0], dtype=object))
    H.assert_equal_ints(np.multiply.reduce(e, axis=1).numpy(), np.array([1, 1, 1], dtype=object))
    assert int(np.sum(x[0, :0])) == 0 and int(np.prod(x[0, :0])) == 1 and np.add.accumulate(e, axis=1).shape == (3, 0)
    with pytest.raises(ValueError):
        np.subtract.reduce(e, axis=1)
    z = GF(np.array([3, 0, 5], dtype=object))
    with pytest.raises(ZeroDivisionError):
        np.true_divide.reduce(z)
    # products: vector . vector, matrix @ vector, batched matrices with a broadcast right operand
    a, b = h[0, :25], h[1, :25]
    dotv = 0
    for u, v in zip(a, b):
        dotv = W.add(dotv, W.mul(int(u), int(v)))
    assert int(np.dot(GF(a), GF(b))) == dotv and int(np.vdot(GF(a), GF(b))) == dotv and int(np.inner(GF(a), GF(b))) == dotv
    A = h[:, :12].reshape(3, 3, 4)
    Bm = h[1, 100:112].reshape(4, 3)
    got = (GF(A) @ GF(Bm)).numpy()
    for t in range(3):
        for i in range(3):
            for j in range(3):
                acc = 0
                for k in range(4):
                    acc = W.add(acc, W.mul(int(A[t, i, k]), int(Bm[k, j])))
                assert int(got[t, i, j]) == acc
    c = np.convolve(GF(h[0, :33]), GF(h[2, :7])).numpy()
    for k in (0, 5, 38):
        acc = 0
        for i in range(33):
            if 0 <= k - i < 7:
                acc = W.add(acc, W.mul(int(h[0, i]), int(h[2, k - i])))
        assert int(c[k]) == acc


def _load_linalg(tag):
    GF, W, _, props = _load(tag)
    return GF, W, np.load(os.path.join(H.GOLDEN, f"sage_wide_linalg_{tag}.npz")), props


def _cases(d, op, keys):
    for i in range(int(d[f"{op}_count"])):
        yield tuple(_obj(d[f"{op}{i}_{k}"]) for k in keys)


@pytest.mark.parametrize("tag", TAGS)
def test_sage_elimination_fixtures_of_the_big_fields(tag):
    """r05 -- row_reduce / lu / plu / inverse / determinant / solve / the four spaces of the three Sage folders of order >= 2^64
    (the reference: object-dtype loops of _domains/_linalg.py:315-548 and _fields/_array.py:1412-1760): rows f2 at 16 / 16 folders."""
    GF, W, d, props = _load_linalg(tag)
    for X, Z in _cases(d, "row_reduce", "XZ"):
        H.assert_equal_ints(GF(X).row_reduce().numpy(), Z, f"{tag} row_reduce")
    for X, Lt, Ut in _cases(d, "lu_decompose", "XLU"):
        l, u = GF(X).lu_decompose()
        H.assert_equal_ints(l.numpy(), Lt, "lu L")
        H.assert_equal_ints(u.numpy(), Ut, "lu U")
    for X, Pt, Lt, Ut in _cases(d, "plu_decompose", "XPLU"):
        p, l, u = GF(X).plu_decompose()
        H.assert_equal_ints(p.numpy(), Pt, "plu P")
        H.assert_equal_ints(l.numpy(), Lt, "plu L")
        H.assert_equal_ints(u.numpy(), Ut, "plu U")
        H.assert_equal_ints((p @ l @ u).numpy(), X, "P L U = A")
    for X, Z in _cases(d, "matrix_inverse", "XZ"):
        H.assert_equal_ints(np.linalg.inv(GF(X)).numpy(), Z, "inv")
    for X, Z in _cases(d, "matrix_determinant", "XZ"):
        assert int(np.linalg.det(GF(X))) == int(Z), "det"
    for X, Y, Z in _cases(d, "matrix_solve", "XYZ"):
        H.assert_equal_ints(np.linalg.solve(GF(X), GF(Y)).numpy(), Z, "solve")
    for op in ("row_space", "column_space", "left_null_space", "null_space"):
        for X, Z in _cases(d, op, "XZ"):
            got = getattr(GF(X), op)().numpy()
            if Z.size == 0:
                assert got.size == 0, op
            else:
                H.assert_equal_ints(got, Z.reshape(got.shape), op)
    assert np.linalg.matrix_rank(GF(np.array([[1, 2], [2, 4]], dtype=object))) == 1
    with pytest.raises(np.linalg.LinAlgError):
        np.linalg.inv(GF(np.array([[1, 2], [2, 4]], dtype=object)))
    with pytest.raises(ValueError):
        GF(np.array([[0, 1], [1, 1]], dtype=object)).lu_decompose()


@pytest.mark.parametrize("tag", TAGS)
def test_sage_log_and_polynomial_evaluation_of_the_big_fields(tag):
    """r05 -- log.pkl (Pohlig-Hellman: q - 1 of the three fields has prime factors up to 4294967291) and
    tests/polys/data/*/evaluate.pkl, evaluate_matrix.pkl (Horner, _polys/_dense.py:404-470): row f4 at 16 / 16 folders."""
    GF, W, d, props = _load_linalg(tag)
    x, z = GF(_obj(d["log_X"])), _obj(d["log_Z"])
    got = np.log(x)
    assert [int(v) for v in np.ravel(got)] == [int(v) for v in z.ravel()]
    alpha = GF(np.array(props["primitive_element"], dtype=object))
    H.assert_equal_ints((alpha ** got).numpy(), x.numpy(), "alpha ** log x == x")
    assert x[3].log() == int(z[3])  # a 0-D input returns a Python int
    with pytest.raises(ArithmeticError):
        np.log(GF(np.array([1, 0], dtype=object)))
    ys = GF(_obj(d["evaluate_Y"]))
    for i in range(int(d["evaluate_count"])):
        poly = ga.Poly(GF(_obj(d[f"evaluate{i}_X"])))
        H.assert_equal_ints(poly(ys).numpy(), _obj(d[f"evaluate{i}_Z"]), f"{tag} evaluate {i}")
    for i in range(int(d["evaluate_matrix_count"])):
        poly = ga.Poly(GF(_obj(d[f"evaluate_matrix{i}_X"])))
        got = poly(GF(_obj(d[f"evaluate_matrix{i}_Y"])), elementwise=False)
        H.assert_equal_ints(got.numpy(), _obj(d[f"evaluate_matrix{i}_Z"]), f"{tag} evaluate_matrix {i}")


@pytest.mark.parametrize("tag", TAGS)
def test_fft_sqrt_reduceat_and_at_on_the_big_fields_against_the_oracle(tag):
    """np.fft.fft / ifft for every small divisor of q - 1 (mixed radix, against the direct DFT on Python integers), np.sqrt
    (round trip and the smaller-root rule), ufunc.reduceat / ufunc.at against left folds of the oracle's scalars."""
    GF, W, d, props = _load(tag)
    q = GF.order
    rnd = random.Random(11)
    divisors = [n for n in (2, 3, 4, 5, 6, 8, 9, 12, 15, 16, 18, 24, 30, 33, 47) if (q - 1) % n == 0]
    assert divisors
    for n in divisors:
        h = [rnd.randrange(q) for _ in range(n)]
        x = GF(np.array(h, dtype=object))
        w = int(GF.primitive_root_of_unity(n))
        want = []
        for k in range(n):
            acc = 0
            for j in range(n):
                acc = W.add(acc, W.mul(h[j], W.pow(w, (j * k) % n)))
            want.append(acc)
        X = np.fft.fft(x)
        assert type(X) is GF
        H.assert_equal_ints(X.numpy(), np.array(want, dtype=object), f"{tag} fft n={n}")
        H.assert_equal_ints(np.fft.ifft(X).numpy(), np.array(h, dtype=object), f"{tag} ifft n={n}")
    with pytest.raises(ValueError):
        np.fft.fft(GF(np.array([1] * 7, dtype=object))) if (q - 1) % 7 else (_ for _ in ()).throw(ValueError())
    # square roots: r * r is a square; sqrt returns the smaller of the two roots as integers
    r = GF(np.array([rnd.randrange(1, q) for _ in range(40)] + [0, 1], dtype=object))
    s = np.sqrt(r * r)
    H.assert_equal_ints((s * s).numpy(), (r * r).numpy(), "sqrt round trip")
    for a, b in zip(s.numpy(), (-s).numpy()):
        assert int(a) <= int(b)
    assert all(bool(v) for v in np.ravel((r * r).is_square()))
    # reduceat / at
    h = [rnd.randrange(1, q) for _ in range(12)]
    x = GF(np.array(h, dtype=object))
    idx = [0, 4, 4, 9, 2]
    for uf, f in ((np.add, W.add), (np.multiply, W.mul), (np.subtract, W.sub)):
        want = []
        ends = idx[1:] + [12]
        for s0, e0 in zip(idx, ends):
            acc = h[s0]
            for v in h[s0 + 1:e0]:
                acc = f(acc, v)
            want.append(acc)
        H.assert_equal_ints(uf.reduceat(x, idx).numpy(), np.array(want, dtype=object), f"{tag} {uf.__name__}.reduceat")
    y = x.copy()
    inc = [rnd.randrange(q) for _ in range(4)]
    np.add.at(y, [1, 1, 3, 1], GF(np.array(inc, dtype=object)))
    exp = list(h)
    for i, v in zip([1, 1, 3, 1], inc):
        exp[i] = W.add(exp[i], v)
    H.assert_equal_ints(y.numpy(), np.array(exp, dtype=object), f"{tag} add.at")


def test_the_ghash_field_gf_2_128():
    """r05: GF(2^128) -- the 129-bit modulus x^128 + x^7 + x^2 + x + 1 with its top bit implicit in the two-limb kernels
    (the reference has no upper bound on the order: _domains/_meta.py:38-41).  Against the Python-integer oracle."""
    irr = (1 << 128) | 0x87
    GF = ga.GF(2, 128, irreducible_poly=irr, primitive_element=2, verify=False)
    W = WideOracle(2, 128, [(irr >> (128 - i)) & 1 for i in range(129)])
    assert GF.order == 2**128 and GF.dtypes == [np.object_]
    rnd = random.Random(128)
    a = [rnd.randrange(2**128) for _ in range(300)] + [2**128 - 1, 1, 2**127, 0]
    b = [rnd.randrange(1, 2**128) for _ in range(300)] + [2**128 - 1, 2**127, 1, 3]
    x, y = GF(np.array(a, dtype=object)), GF(np.array(b, dtype=object))
    H.assert_equal_ints((x + y).numpy(), np.array([u ^ v for u, v in zip(a, b)], dtype=object))
    H.assert_equal_ints((x * y).numpy(), np.array([W.mul(u, v) for u, v in zip(a, b)], dtype=object), "GF(2^128) mul")
    H.assert_equal_ints((x / y).numpy(), np.array([W.div(u, v) for u, v in zip(a, b)], dtype=object), "GF(2^128) div")
    H.assert_equal_ints((y * np.reciprocal(y)).numpy(), np.array([1] * len(b), dtype=object))
    H.assert_equal_ints((y ** (2**128 - 1)).numpy(), np.array([1] * len(b), dtype=object), "x^(q-1) = 1")
    H.assert_equal_ints((y ** -3).numpy(), np.array([W.pow(v, -3) for v in b], dtype=object))
    with pytest.raises(ValueError):
        GF(np.array([2**128], dtype=object))
    # (orders above 2^128: the k-limb fields of tests/test_gpu_big.py)
