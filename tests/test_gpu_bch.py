"""Parity of the BCH path (the RS decoder kernel with base field != extension field)
against the reference's Sage fixtures, outputs of the reference itself, and the oracle.  Bit-exact."""
import json

import numpy as np
import pytest

import galois_amd as ga
from oracle import gf_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_sage_bch_fixtures():
    """tests/codes/test_bch.py:105-170 over all 204 fixtures: encode (matrix, vector, parity, shortened), detect, and a
    decode round trip with <= t errors (tests/codes/conftest.py:174-228)."""
    names, d = H.sage_bch()
    rng = np.random.default_rng(11)
    for key in names:
        meta = json.loads(str(d[f"{key}/meta"]))
        p = meta["q"]
        bch = ga.BCH(meta["n"], meta["k"], d=meta["d"], field=ga.GF(p), alpha=meta["alpha"], c=meta["c"],
                     systematic=meta["is_systematic"])
        msgs = d[f"{key}/messages"].astype(np.int64)
        cw = bch.encode(msgs)
        assert type(cw) is bch.field
        H.assert_equal_ints(cw.numpy(), d[f"{key}/codewords"], key)
        H.assert_equal_ints(bch.encode(msgs[0]).numpy(), d[f"{key}/codewords"][0], key + " 1-D")
        if meta["is_systematic"]:
            H.assert_equal_ints(bch.encode(msgs, output="parity").numpy(), d[f"{key}/codewords"][:, meta["k"]:], key + " parity")
        else:
            with pytest.raises(ValueError):
                bch.encode(msgs, output="parity")
        assert not bch.detect(cw).any(), key
        for short in (False, True):
            if short and f"{key}/short_messages" not in d:
                continue
            M = d[f"{key}/short_messages" if short else f"{key}/messages"].astype(np.int64)
            C = d[f"{key}/short_codewords" if short else f"{key}/codewords"].astype(np.int64)
            if short:
                H.assert_equal_ints(bch.encode(M).numpy(), C, key + " shortened")
            ns = C.shape[1]
            R = C.copy()
            ne = rng.integers(0, bch.t + 1, R.shape[0])
            for i in range(R.shape[0]):
                pos = rng.choice(ns, min(ne[i], ns), replace=False)
                R[i, pos] = (R[i, pos] + rng.integers(1, p, pos.size)) % p
                ne[i] = pos.size
            dec, nerr = bch.decode(R, errors=True)
            H.assert_equal_ints(dec.numpy(), M, key + " decode")
            assert np.array_equal(nerr, ne), key
            assert np.array_equal(bch.detect(R), ne > 0), key


def _check_device_bch(d, tag):
    meta = json.loads(str(d[f"bch/{tag}/meta"]))
    p = meta["p"]
    ext = ga.GF(p, meta["ext_m"], irreducible_poly=meta["ext_irr"], primitive_element=meta["ext_alpha"])
    bch = ga.BCH(meta["n"], meta["k"], d=meta["d"], field=ga.GF(p), extension_field=ext, alpha=meta["alpha"], c=meta["c"],
                 systematic=meta["systematic"])
    assert (bch.k, bch.d) == (meta["k"], meta["d"])
    H.assert_equal_ints(bch.generator_poly.coeffs, d[f"bch/{tag}/generator_poly"])
    H.assert_equal_ints(bch.roots, d[f"bch/{tag}/roots"])
    M = d[f"bch/{tag}/messages"].astype(np.int64)
    H.assert_equal_ints(bch.encode(M).numpy(), d[f"bch/{tag}/codewords"], tag + " encode")
    R, E, raises = d[f"bch/{tag}/received"].astype(np.int64), d[f"bch/{tag}/erasures"], d[f"bch/{tag}/raises"]
    assert np.array_equal(bch.detect(R), d[f"bch/{tag}/detected"])
    ok = ~raises
    dec, nerr = bch.decode(R[ok], erasures=E[ok], output="codeword", errors=True)
    H.assert_equal_ints(dec.numpy(), d[f"bch/{tag}/decoded"][ok], tag + " decoded codewords")
    assert np.array_equal(nerr, d[f"bch/{tag}/n_errors"][ok])
    H.assert_equal_ints(bch.decode(R[ok], erasures=E[ok]).numpy(), d[f"bch/{tag}/decoded_message"][ok], tag + " messages")
    for i in np.nonzero(raises)[0]:
        with pytest.raises(ValueError):
            bch.decode(R[i], erasures=E[i])
    if raises.any():
        with pytest.raises(ValueError):
            bch.decode(R, erasures=E)


@pytest.mark.parametrize("tag", ["bch15_7", "bch15_5_c3", "bch31_16", "bch63_45", "bch63_36_short", "bch255_223",
                                 "bch127_99_nonsys", "bch13_4_gf3", "bch26_14_gf3", "bch26_8_gf3_c3", "bch80_60_gf3",
                                 "bch24_gf5", "bch26_14_gf3_nonsys_short"])
def test_reference_generated_cases(tag):
    """Errors, erasures, beyond-capacity words and the rows on which the reference raises (symbols outside GF(p))."""
    _check_device_bch(H.reference_bch_outputs(), tag)


@pytest.mark.parametrize("tag", H.WIDE_BCH_CASES)
def test_wide_reference_generated_cases(tag):
    """BCH codes of length 511, 728 (GF(3)) and 1023 -- syndrome fields GF(2^9), GF(3^6), GF(2^10) -- against outputs of the
    reference itself (gfa_rs_wide.hip)."""
    _check_device_bch(H.reference_wide_codes(), tag)


@pytest.mark.parametrize("p,n,d,c,systematic", [(2, 255, 9, 1, True), (2, 255, 37, 1, True), (2, 127, 21, 0, True),
                                                (2, 63, 13, 1, False), (3, 80, 9, 1, True), (3, 242, 11, 2, True),
                                                (5, 124, 9, 1, True), (7, 48, 7, 1, False),
                                                (2, 1023, 21, 1, True), (2, 511, 9, 0, False), (3, 728, 9, 1, True),
                                                (2, 4095, 7, 1, True), (2, 2047, 9, 1, True), (5, 624, 7, 2, True)])
def test_random_batches_against_oracle(p, n, d, c, systematic):
    """Larger batches with 0 .. t+2 errors and random erasures: decoded rows, n_errors and the out-of-field rows match
    the oracle row by row (the batch call raises iff any row leaves GF(p))."""
    bch = ga.BCH(n, d=d, field=ga.GF(p), c=c, systematic=systematic)
    k = bch.k
    ext = bch.extension_field
    F = O.OracleField(p, ext.degree, int(ext.irreducible_poly), int(ext.primitive_element), lookup=True)
    B = O.OracleBCH(F, n, d=d, alpha=bch.alpha, c=c, systematic=systematic)
    assert (B.k, B.d, B.generator_poly) == (bch.k, bch.d, [int(v) for v in bch.generator_poly.coeffs])
    rng = np.random.default_rng(n * 1000 + k)
    N = 512
    M = rng.integers(0, p, (N, bch.k))
    C = bch.encode(M).numpy().astype(np.int64)
    H.assert_equal_ints(C, B.encode(M))
    R = C.copy()
    E = np.zeros((N, n), dtype=bool)
    for i in range(N):
        ne = int(rng.integers(0, bch.t + 3))
        pos = rng.choice(n, ne, replace=False)
        R[i, pos] = (R[i, pos] + rng.integers(1, p, ne)) % p
        if i % 3 == 0:
            epos = rng.choice(n, int(rng.integers(0, bch.d)), replace=False)
            E[i, epos] = True
    odec, onerr = B.decode(R, E)
    bad = ((odec < 0) | (odec >= p)).any(axis=1)
    ok = ~bad
    dec, nerr = bch.decode(R[ok], erasures=E[ok], output="codeword", errors=True)
    assert np.array_equal(nerr, onerr[ok])
    H.assert_equal_ints(dec.numpy(), odec[ok])
    H.assert_equal_ints(bch.decode(R[ok], erasures=E[ok]).numpy(), B.message_of(odec[ok]))
    assert np.array_equal(bch.detect(R), B.detect(R))
    for i in np.nonzero(bad)[0][:8]:
        with pytest.raises(ValueError):
            bch.decode(R[i], erasures=E[i])


def test_bch_255_full_batch_round_trip():
    """2^18 codewords of binary BCH(255, 223), t = 4: encode -> <= t errors -> decode returns the messages."""
    bch = ga.BCH(255, 223)
    rng = np.random.default_rng(3)
    N = 1 << 18
    M = rng.integers(0, 2, (N, 223), dtype=np.uint8)
    C = bch.encode(M)
    R = C.numpy().copy()
    ne = rng.integers(0, 5, N)
    cols = np.argsort(rng.random((N, 255)), axis=1)[:, :4]
    flip = np.arange(4)[None, :] < ne[:, None]
    rows = np.repeat(np.arange(N)[:, None], 4, axis=1)
    R[rows[flip], cols[flip]] ^= 1
    dec, nerr = bch.decode(R, errors=True)
    assert np.array_equal(dec.numpy(), M)
    assert np.array_equal(nerr, ne)


def test_front_end_errors():
    """tests/codes/test_bch.py:20-46, 97-103, 144-150."""
    with pytest.raises(TypeError):
        ga.BCH(15.0, 7)
    with pytest.raises(TypeError):
        ga.BCH(15, 7.0)
    with pytest.raises(TypeError):
        ga.BCH(15, 7, field=2)
    with pytest.raises(TypeError):
        ga.BCH(15, 7, c=1.0)
    with pytest.raises(ValueError):
        ga.BCH(15, 7, d=0)
    with pytest.raises(ValueError):
        ga.BCH(15, 7, c=-1)
    with pytest.raises(ValueError):
        ga.BCH(15, 12)
    with pytest.raises(ValueError):
        ga.BCH(15, 7, field=ga.GF(2**2))
    with pytest.raises(ValueError):
        ga.BCH(15)
    with pytest.raises(ValueError):
        ga.BCH(15, 7, d=7)
    bch = ga.BCH(15, 7)
    with pytest.raises(ValueError):
        bch.encode(np.zeros(8, dtype=int))
    with pytest.raises(ValueError):
        bch.decode(np.zeros(16, dtype=int))
    with pytest.raises(ValueError):
        ga.BCH(15, 7, systematic=False).encode(np.zeros(7, dtype=int), output="parity")


@pytest.mark.parametrize("p,n,d,c,nu", [(2, 4095, 3, 3, 0), (3, 728, 2, 3, 1), (2, 1023, 5, 1, 2), (2, 511, 3, 1, 2)])
def test_wide_miscorrections_stay_out_of_field(p, n, d, c, nu):
    """Beyond-capacity words of codes with a large syndrome field: the Forney values of a miscorrection are arbitrary elements
    of GF(p^m) (up to 4095 here), so the corrected symbol leaves GF(p) and the reference raises (_bch.py:1300).  The device
    stores symbols as uint8: the out-of-field value must survive the narrowing (found by tools/fuzz_codes.py ... wide: a plain
    truncation wrapped about 1 in 100 of them back into range).  nu erasures per word on top of the errors."""
    bch = ga.BCH(n, d=d, field=ga.GF(p), c=c)
    ext = bch.extension_field
    F = O.OracleField(p, ext.degree, int(ext.irreducible_poly), int(ext.primitive_element), lookup=True)
    B = O.OracleBCH(F, n, d=d, alpha=bch.alpha, c=c)
    rng = np.random.default_rng(n + d)
    N = 600
    C = bch.encode(rng.integers(0, p, (N, bch.k))).numpy().astype(np.int64)
    R = C.copy()
    E = np.zeros((N, n), dtype=bool)
    for i in range(N):
        pos = rng.choice(n, bch.t + 1 + int(rng.integers(0, 3)) + nu, replace=False)
        err = pos[nu:]
        R[i, err] = (R[i, err] + rng.integers(1, p, err.size)) % p
        E[i, pos[:nu]] = True
    odec, onerr = B.decode(R, E if nu else None)
    bad = ((odec < 0) | (odec >= p)).any(axis=1)
    assert bad.sum() > 10, bad.sum()
    for i in np.nonzero(bad)[0]:
        with pytest.raises(ValueError):
            bch.decode(R[i], erasures=E[i] if nu else None)
    ok = ~bad
    dec, nerr = bch.decode(R[ok], erasures=E[ok] if nu else None, output="codeword", errors=True)  # possibly an empty batch
    assert np.array_equal(nerr, onerr[ok]) and np.array_equal(dec.numpy().astype(np.int64).reshape(-1, n), odec[ok])
