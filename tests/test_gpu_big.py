"""Fields of order above 2^128 on the device (r05: galois_amd/_big.py, csrc/gfa_big.hip; 4 / 8 / 16 limbs of 64 bits) -- the
reference has no upper bound on the order (src/galois/_domains/_meta.py:38-41: dtype=object Python integers through the
python-calculate ufuncs).  Every kind (prime, binary, extension) and every limb count against oracle/wide_oracle.py, the
restatement of the reference's scalar formulas on Python integers (pinned to the reference's Sage vectors of its three biggest
fields by tests/test_oracle_golden.py)."""
import random

import numpy as np
import pytest

import galois_amd as ga
from oracle.wide_oracle import WideOracle
from tests import helpers as H

pytestmark = pytest.mark.gpu

K163 = (1 << 163) | (1 << 7) | (1 << 6) | (1 << 3) | 1
B409 = (1 << 409) | (1 << 87) | 1
FIELDS = {
    "p25519": dict(p=2**255 - 19, m=1, irr=None, alpha=2, nl=4),
    "gf2e163": dict(p=2, m=163, irr=K163, alpha=2, nl=4),
    "gf65537e12": dict(p=65537, m=12, irr=[1] + [0] * 10 + [1, 2], alpha=65537, nl=4),
    "gf2e409": dict(p=2, m=409, irr=B409, alpha=2, nl=8),
    "p521": dict(p=2**521 - 1, m=1, irr=None, alpha=3, nl=16),
    "gf2e256": dict(p=2, m=256, irr=(1 << 256) | (1 << 10) | (1 << 5) | (1 << 2) | 1, alpha=2, nl=4),  # m = 64 nl: implicit top bit
}


def _make(tag):
    f = FIELDS[tag]
    p, m = f["p"], f["m"]
    if m == 1:
        GF = ga.GF(p, primitive_element=f["alpha"], verify=False)
        return GF, WideOracle(p, 1, None), f
    irr = f["irr"]
    coeffs = irr if isinstance(irr, list) else [(irr >> (m - i)) & 1 for i in range(m + 1)]
    GF = ga.GF(p, m, irreducible_poly=irr, primitive_element=f["alpha"], verify=False)
    return GF, WideOracle(p, m, coeffs), f


def _obj(v):
    return np.array([int(x) for x in v], dtype=object)


@pytest.mark.parametrize("tag", list(FIELDS))
def test_elementwise_ufuncs_of_the_k_limb_fields_against_the_oracle(tag):
    GF, W, f = _make(tag)
    q, p = GF.order, GF.characteristic
    assert GF._NL == f["nl"] and GF.dtypes == [np.object_] and GF.order == p ** f["m"]
    rng = random.Random(len(tag))
    edge = [0, 1, 2, q - 1, q - 2, p - 1 if p < q else q - 1, (1 << 64) - 1, 1 << 64, (1 << 128) + 1, (1 << 192) - 1, q // 2, q // 3]
    edge = [v % q for v in edge]
    a = edge + [rng.randrange(q) for _ in range(150)]
    b = [rng.choice(edge) for _ in range(len(edge))] + [rng.randrange(q) for _ in range(150)]
    A, B = GF(_obj(a)), GF(_obj(b))
    assert type(A + B) is GF and (A + B).dtype == np.dtype(object)
    H.assert_equal_ints((A + B).numpy(), _obj(W.add(x, y) for x, y in zip(a, b)), f"{tag} add")
    H.assert_equal_ints((A - B).numpy(), _obj(W.sub(x, y) for x, y in zip(a, b)), f"{tag} sub")
    H.assert_equal_ints((A * B).numpy(), _obj(W.mul(x, y) for x, y in zip(a, b)), f"{tag} mul")
    H.assert_equal_ints((-A).numpy(), _obj(W.neg(x) for x in a), f"{tag} neg")
    bnz = [y if y else 1 for y in b]
    Bnz = GF(_obj(bnz))
    n_div = 40 if f["nl"] > 4 or f["m"] > 1 else len(a)  # an inversion is an exponentiation by q - 2: keep the oracle's share short
    H.assert_equal_ints((A[:n_div] / Bnz[:n_div]).numpy(), _obj(W.div(x, y) for x, y in zip(a[:n_div], bnz[:n_div])), f"{tag} div")
    H.assert_equal_ints(np.reciprocal(Bnz[:n_div]).numpy(), _obj(W.inv(y) for y in bnz[:n_div]), f"{tag} reciprocal")
    H.assert_equal_ints((Bnz[:n_div] * np.reciprocal(Bnz[:n_div])).numpy(), _obj([1] * n_div))
    es = [0, 1, 2, -1, -2, q - 1, q, -(q - 1), 3 * q + 5] + [rng.randrange(-q, q) for _ in range(12)]
    sub = [x if x else 3 for x in a[:len(es)]]
    got = GF(_obj(sub)) ** np.array(es, dtype=object)
    H.assert_equal_ints(got.numpy(), _obj(W.pow(x, e) for x, e in zip(sub, es)), f"{tag} power")
    assert int(GF(0) ** 0) == 1 and int(GF(0) ** 5) == 0 and int(np.square(GF(q - 1))) == W.mul(q - 1, q - 1)
    k = 12345678901234567890123456789
    H.assert_equal_ints((A * k).numpy(), _obj(W.mul(x, k % p) for x in a), f"{tag} field * integer")
    # broadcasting against a scalar; the reference's error behaviour
    H.assert_equal_ints((A * GF(7 % q)).numpy(), _obj(W.mul(x, 7 % q) for x in a))
    with pytest.raises(ZeroDivisionError):
        A / GF(_obj([0] * len(a)))
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


@pytest.mark.parametrize("tag", ["p25519", "gf2e163", "gf65537e12", "gf2e409"])
def test_folds_and_array_surface_of_the_k_limb_fields(tag):
    GF, W, f = _make(tag)
    q = GF.order
    rng = random.Random(5)
    h = np.array([rng.randrange(1, q) for _ in range(24)], dtype=object).reshape(4, 6)
    x = GF(h)
    assert x.shape == (4, 6) and x.ndim == 2 and x.size == 24 and len(x) == 4
    H.assert_equal_ints(x.numpy(), h)
    H.assert_equal_ints(x.T.numpy(), h.T)
    H.assert_equal_ints(x[1:3, ::2].numpy(), h[1:3, ::2])
    H.assert_equal_ints(x.reshape(6, 4).numpy(), h.reshape(6, 4))
    H.assert_equal_ints(np.transpose(x).numpy(), h.T)
    H.assert_equal_ints(np.concatenate([x, x], axis=1).numpy(), np.concatenate([h, h], axis=1))
    H.assert_equal_ints(np.stack([x[0], x[1]]).numpy(), np.stack([h[0], h[1]]))
    m = rng.random() < 2  # noqa: F841
    cond = np.array([[bool((i + j) % 2) for j in range(6)] for i in range(4)])
    H.assert_equal_ints(np.where(cond, x, GF.Zeros((4, 6))).numpy(), np.where(cond, h, 0))
    y = x.copy()
    y[0, 0] = 5
    assert int(y[0, 0]) == 5 and int(x[0, 0]) == int(h[0, 0]) and not np.array_equal(x, y) and np.array_equal(x, x.copy())
    assert (x == x).all() and bool((x == GF(h))[2, 3])
    fold = {np.add: W.add, np.subtract: W.sub, np.multiply: W.mul}

    def left_fold(fn, row):
        acc = int(row[0])
        out = [acc]
        for v in row[1:]:
            acc = fn(acc, int(v))
            out.append(acc)
        return out

    for uf, fn in fold.items():
        H.assert_equal_ints(uf.reduce(x, axis=1).numpy(), _obj(left_fold(fn, r)[-1] for r in h), f"{tag} {uf.__name__}.reduce")
        H.assert_equal_ints(uf.reduce(x, axis=0).numpy(), _obj(left_fold(fn, c)[-1] for c in h.T))
        H.assert_equal_ints(uf.accumulate(x, axis=1).numpy(), np.array([left_fold(fn, r) for r in h], dtype=object))
    assert int(np.sum(x)) == left_fold(W.add, h.ravel())[-1] and int(np.prod(x[0])) == left_fold(W.mul, h[0])[-1]
    assert int(np.add.reduce(x[0], initial=GF(9))) == W.add(9, left_fold(W.add, h[0])[-1])
    # NumPy's fold from a seed: ((9 - x0) - x1) ... = 9 - (x0 + x1 + ...)  (the dual operation folds the body)
    want = 9
    for v in h[0]:
        want = W.sub(want, int(v))
    assert int(np.subtract.reduce(x[0], initial=GF(9))) == want
    assert int(np.add.reduce(x[0, :0], initial=GF(9))) == 9
    assert np.sum(x, axis=1, keepdims=True).shape == (4, 1)
    H.assert_equal_ints(np.multiply.outer(x[0, :3], x[1, :2]).numpy(), np.array([[W.mul(int(u), int(v)) for v in h[1, :2]] for u in h[0, :3]], dtype=object))
    # compositions of the element-wise kernels: where= on calls, np.convolve, @
    old = GF(h[::-1].copy())
    got = np.multiply(x, x, where=cond, out=old)
    H.assert_equal_ints(got.numpy(), np.where(cond, np.array([[W.mul(int(v), int(v)) for v in r] for r in h], dtype=object), h[::-1]), f"{tag} where=")
    cv = np.convolve(x[0], x[1, :4])
    want = [0] * 9
    for i, u in enumerate(h[0]):
        for j, v in enumerate(h[1, :4]):
            want[i + j] = W.add(want[i + j], W.mul(int(u), int(v)))
    H.assert_equal_ints(cv.numpy(), _obj(want), f"{tag} convolve")
    mm = x @ x.T
    wm = [[0] * 4 for _ in range(4)]
    for i in range(4):
        for j in range(4):
            acc = 0
            for k in range(6):
                acc = W.add(acc, W.mul(int(h[i, k]), int(h[j, k])))
            wm[i][j] = acc
    H.assert_equal_ints(mm.numpy(), np.array(wm, dtype=object), f"{tag} matmul")
    H.assert_equal_ints((x @ x[0]).numpy(), _obj(wm[i][0] for i in range(4)))
    for bad in (lambda: np.add.reduce(x, where=cond, initial=0), lambda: np.log(x), lambda: np.sqrt(x), lambda: np.sort(x), lambda: np.linalg.inv(mm)):
        with pytest.raises(NotImplementedError):
            bad()


def test_factory_rules_above_2_128():
    """No default generator above 2^128 (finding one factors q - 1; the reference's own default search would not return): the
    caller names the polynomial and the primitive element and switches verification off.  1024 bits is the limit."""
    p = 2**255 - 19
    with pytest.raises(ValueError, match="primitive_element"):
        ga.GF(p)
    with pytest.raises(ValueError, match="verify=False"):
        ga.GF(p, primitive_element=2)
    with pytest.raises(NotImplementedError, match="1024"):
        ga.GF(2, 1100, irreducible_poly=(1 << 1100) | 3, primitive_element=2, verify=False)
    assert ga.GF(p, primitive_element=2, verify=False) is ga.GF(p, primitive_element=2, verify=False)
