"""Parity of the Reed-Solomon kernels against golden vectors and the oracle.  Bit-exact."""
import json

import numpy as np
import pytest

import galois_amd as ga
from oracle import gf_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_sage_encode_fixtures():
    """tests/codes/test_reed_solomon.py:98-133: all 56 Sage fixtures, systematic and non-systematic."""
    names, d = H.sage_rs()
    n_checked = 0
    for key in names:
        meta = json.loads(str(d[f"{key}/meta"]))
        rs = ga.ReedSolomon(meta["n"], meta["k"], field=ga.GF(meta["q"]), alpha=meta["alpha"], c=meta["c"],
                            systematic=meta["is_systematic"])
        msgs = d[f"{key}/messages"].astype(np.int64)
        cw = rs.encode(msgs)
        assert type(cw) is rs.field
        H.assert_equal_ints(cw.numpy(), d[f"{key}/codewords"], key)
        H.assert_equal_ints(rs.encode(msgs[0]).numpy(), d[f"{key}/codewords"][0], key + " 1-D")
        if meta["is_systematic"]:
            H.assert_equal_ints(rs.encode(msgs, output="parity").numpy(), d[f"{key}/codewords"][:, meta["k"]:], key + " parity")
        else:
            with pytest.raises(ValueError):
                rs.encode(msgs, output="parity")
        H.assert_equal_ints(rs.encode(msgs.tolist()).numpy(), d[f"{key}/codewords"], key + " list input")
        if f"{key}/short_messages" in d:
            H.assert_equal_ints(rs.encode(d[f"{key}/short_messages"].astype(np.int64)).numpy(), d[f"{key}/short_codewords"], key + " shortened")
        assert not rs.detect(cw).any()
        # round trip with <= t errors (tests/codes/conftest.py:174-228)
        rng = np.random.default_rng(7)
        R = d[f"{key}/codewords"].astype(np.int64).copy()
        ne = rng.integers(0, rs.t + 1, R.shape[0])
        for i in range(R.shape[0]):
            pos = rng.choice(meta["n"], ne[i], replace=False)
            R[i, pos] = (R[i, pos] + rng.integers(1, meta["q"], ne[i])) % meta["q"]
        dec, nerr = rs.decode(R, errors=True)
        H.assert_equal_ints(dec.numpy(), msgs, key + " decode")
        assert np.array_equal(nerr, ne)
        n_checked += 1
    assert n_checked == 56


def _check_device_rs(d, tag):
    meta = json.loads(str(d[f"rs/{tag}/meta"]))
    GF = ga.GF(meta["p"], meta["m"], irreducible_poly=meta["irr"], primitive_element=meta["field_alpha"]) if meta["m"] > 1 \
        else ga.GF(meta["p"], primitive_element=meta["field_alpha"])
    rs = ga.ReedSolomon(meta["n"], meta["k"], field=GF, c=meta["c"], alpha=meta["alpha"])
    H.assert_equal_ints(rs.generator_poly.coeffs, d[f"rs/{tag}/generator_poly"])
    M = d[f"rs/{tag}/messages"].astype(np.int64)
    H.assert_equal_ints(rs.encode(M).numpy(), d[f"rs/{tag}/codewords"], tag + " encode")
    R, E = d[f"rs/{tag}/received"].astype(np.int64), d[f"rs/{tag}/erasures"]
    dec, nerr = rs.decode(R, erasures=E, output="codeword", errors=True)
    assert np.array_equal(nerr, d[f"rs/{tag}/n_errors"]), (tag, nerr, d[f"rs/{tag}/n_errors"])
    H.assert_equal_ints(dec.numpy(), d[f"rs/{tag}/decoded"], tag + " decoded")
    assert np.array_equal(rs.detect(R), d[f"rs/{tag}/detected"]), tag + " detect"
    # 1-D forms
    d1, n1 = rs.decode(R[1], erasures=E[1], output="codeword", errors=True)
    assert isinstance(n1, int) and n1 == int(d[f"rs/{tag}/n_errors"][1])
    H.assert_equal_ints(d1.numpy(), d[f"rs/{tag}/decoded"][1])
    ks = M.shape[1]
    H.assert_equal_ints(rs.decode(R[:2], erasures=E[:2]).numpy(), d[f"rs/{tag}/decoded"][:2, :ks])


def test_reference_generated_cases():
    d = H.reference_outputs()
    tags = sorted({k.split("/")[1] for k in d.files if k.startswith("rs/") and k.endswith("/meta")})
    for tag in tags:
        _check_device_rs(d, tag)
    rs = ga.ReedSolomon(255, 223)
    H.assert_equal_ints(rs.encode(np.arange(223), output="parity").numpy(), d["rs/kat_arange_parity"])


@pytest.mark.parametrize("q,n,k,c", [(2**8, 255, 223, 1), (2**8, 255, 239, 0), (2**8, 85, 65, 1), (2**4, 15, 9, 2), (3**4, 80, 60, 1),
                                     (3**3, 26, 20, 1), (31, 30, 22, 1), (2**8, 255, 127, 1), (5**3, 124, 100, 3),
                                     (2**8, 255, 215, 0), (2**8, 255, 203, 1), (2**8, 51, 19, 1)])  # d - 1 = 40, 52 (the wave kernel's 64-slot form, the
                                     # division-form Berlekamp-Massey), and 32 roots over a 51-point subgroup
def test_random_batches_against_oracle(q, n, k, c):
    if q == 2**8 and n == 255:
        rs = ga.ReedSolomon(n, k, c=c)  # default field: GF(2^8) with the Matlab primitive polynomial
    else:
        rs = ga.ReedSolomon(n, k, field=ga.GF(q), c=c)
    GF = rs.field
    p, m = GF.characteristic, GF.degree
    F = O.OracleField(p, m, int(GF.irreducible_poly) if m > 1 else None, GF._primitive_element_int, lookup=True)
    R_ = O.OracleRS(F, n, k, alpha=rs.alpha, c=c)
    rng = np.random.default_rng(n * 1000 + k)
    N = 1500
    for shorten in (0, min(5, k - 1)):
        ks, ns = k - shorten, n - shorten
        M = rng.integers(0, q, (N, ks)).astype(np.uint8)
        C = rs.encode(M).numpy()
        assert np.array_equal(C, R_.encode_u8(M))
        R = C.copy()
        E = np.zeros((N, ns), dtype=bool)
        t = (n - k) // 2
        for i in range(N):
            ne = int(rng.integers(0, t + 3))
            nu = int(rng.integers(0, n - k + 2)) if i % 3 == 0 else 0
            pos = rng.choice(ns, min(ne, ns), replace=False)
            R[i, pos] = (R[i, pos].astype(np.int64) + rng.integers(1, q, pos.size)) % q
            if nu:
                epos = rng.choice(ns, min(nu, ns), replace=False)
                E[i, epos] = True
                R[i, epos] = rng.integers(0, q, epos.size)
        dec, nerr = rs.decode(R, erasures=E, output="codeword", errors=True)
        odec, onerr = R_.decode_u8(R, E)
        assert np.array_equal(nerr, onerr), np.nonzero(nerr != onerr)[0][:10]
        assert np.array_equal(dec.numpy(), odec)
        dec2, nerr2 = rs.decode(R, output="codeword", errors=True)
        odec2, onerr2 = R_.decode_u8(R)
        assert np.array_equal(nerr2, onerr2) and np.array_equal(dec2.numpy(), odec2)
        assert np.array_equal(rs.detect(R), R_.detect(R))


def test_full_size_2e20_codewords_round_trip():
    """BASELINE.json configs[3] at full size (one GPU holds all 2^20 codewords here): encode -> corrupt -> decode."""
    rs = ga.ReedSolomon(255, 223)
    B = 1 << 20
    rng = np.random.default_rng(4)
    M = rng.integers(0, 256, (B, 223), dtype=np.uint8)
    C = rs.encode(M)
    Ch = C.numpy()
    assert np.array_equal(Ch[:, :223], M)
    assert not rs.detect(C).any()
    # (from 2^18 words the encoder is the register-resident LFSR kernel: its parity-only form against the full codewords above)
    assert np.array_equal(rs.encode(M[: 1 << 18], output="parity").numpy(), Ch[: 1 << 18, 223:])
    # errors: e_i ~ U{0..16} at random positions -- vectorised construction (argsort of random keys)
    ne = rng.integers(0, 17, B)
    keys = rng.random((B, 255), dtype=np.float32)
    order = np.argsort(keys, axis=1)[:, :16]
    R = Ch.copy()
    mask = np.arange(16)[None, :] < ne[:, None]
    rows = np.repeat(np.arange(B), 16).reshape(B, 16)
    vals = rng.integers(1, 256, (B, 16), dtype=np.uint8)
    R[rows[mask], order[mask]] ^= vals[mask]
    dec, nerr = rs.decode(R, output="codeword", errors=True)
    assert np.array_equal(nerr, ne)
    assert np.array_equal(dec.numpy(), Ch)
    assert np.array_equal(rs.detect(R), ne > 0)
    # oracle on a bounded sample
    F = O.OracleField(2, 8, 285, 2, lookup=True)
    OR = O.OracleRS(F, 255, 223)
    sel = rng.choice(B, 512, replace=False)
    assert np.array_equal(OR.encode_u8(M[sel]), Ch[sel])
    od, on = OR.decode_u8(R[sel])
    assert np.array_equal(od, Ch[sel]) and np.array_equal(on, ne[sel])


def test_front_end_errors():
    rs = ga.ReedSolomon(15, 9)
    with pytest.raises(ValueError):
        rs.encode(np.zeros(10, dtype=int))
    with pytest.raises(ValueError):
        rs.encode(np.zeros((2, 2, 9), dtype=int))
    with pytest.raises(ValueError):
        rs.encode([1, 2, 3], output="message")
    with pytest.raises(ValueError):
        rs.decode(np.zeros(6, dtype=int))  # shorter than n-k+1
    with pytest.raises(ValueError):
        rs.decode(np.zeros(15, dtype=int), output="parity")
    with pytest.raises(ValueError):
        rs.encode([16] * 9)
    with pytest.raises(TypeError):
        rs.decode(np.zeros(15, dtype=int), erasures=np.zeros(15, dtype=int))
    with pytest.raises(ValueError):
        rs.decode(np.zeros(15, dtype=int), erasures=np.zeros(14, dtype=bool))
    m, n = rs.decode(np.zeros(15, dtype=int), errors=True)
    assert n == 0 and m.shape == (9,)


def test_c_abi_decode_in_place():
    """gfa_rs_decode with out_codeword == recv (the pre-pass then skips its copy and the wave kernel patches in place)."""
    import ctypes

    import torch

    from galois_amd import _lib as L

    rs = ga.ReedSolomon(255, 223)
    rng = np.random.default_rng(77)
    N = 4096
    M = rng.integers(0, 256, (N, 223), dtype=np.uint8)
    C = rs.encode(M).numpy()
    R = C.copy()
    ne = rng.integers(0, 20, N)
    for i in range(N):
        pos = rng.choice(255, ne[i], replace=False)
        R[i, pos] ^= rng.integers(1, 256, ne[i], dtype=np.uint8)
    F = O.OracleField(2, 8, 285, 2, lookup=True)
    want, wn = O.OracleRS(F, 255, 223).decode_u8(R)
    buf = torch.from_numpy(R).cuda()
    nerr = torch.empty(N, dtype=torch.int64, device="cuda")
    L.check(L.lib().gfa_rs_decode(rs._handle, buf.data_ptr(), None, 255, buf.data_ptr(), nerr.data_ptr(), N, L.U8,
                                  torch.cuda.current_stream().cuda_stream), "gfa_rs_decode")
    assert np.array_equal(nerr.cpu().numpy(), wn)
    assert np.array_equal(buf.cpu().numpy(), want)


def test_identity_code_with_erasures():
    """n == k (d = 1): any erased word fails (u > d - 1, _bch.py:1357-1360), the others pass through (found by tools/fuzz_codes.py)."""
    GF = ga.GF(2**3)
    rs = ga.ReedSolomon(7, 7, field=GF)
    R = np.arange(21).reshape(3, 7) % 8
    E = np.zeros((3, 7), dtype=bool)
    E[1, 2] = True
    dec, nerr = rs.decode(R, erasures=E, output="codeword", errors=True)
    assert list(nerr) == [0, -1, 0] and np.array_equal(dec.numpy(), R)
    dec, nerr = rs.decode(R, errors=True)
    assert list(nerr) == [0, 0, 0] and np.array_equal(dec.numpy(), R)


# ---- codes over fields above 256 elements (gfa_rs_wide.hip) ------------------------------------------------------------
@pytest.mark.parametrize("tag", H.WIDE_RS_CASES)
def test_wide_reference_generated_cases(tag):
    """RS over GF(2^10), GF(2^9), GF(3^6) against outputs of the reference itself (errors, erasures, failures, shortened)."""
    _check_device_rs(H.reference_wide_codes(), tag)


@pytest.mark.parametrize("q,n,k,c,N", [(2**10, 1023, 1003, 1, 300), (2**10, 1023, 901, 0, 60), (3**6, 728, 712, 1, 200),
                                       (2**9, 511, 479, 2, 200), (2**12, 4095, 4063, 1, 40), (2**16, 65535, 65519, 1, 6),
                                       (7**4, 2400, 2380, 1, 40), (2**10, 341, 321, 1, 100)])
def test_wide_random_batches_against_oracle(q, n, k, c, N):
    GF = ga.GF(q)
    rs = ga.ReedSolomon(n, k, field=GF, c=c)
    p, m = GF.characteristic, GF.degree
    F = O.OracleField(p, m, int(GF.irreducible_poly), GF._primitive_element_int, lookup=True)
    R_ = O.OracleRS(F, n, k, alpha=rs.alpha, c=c)
    rng = np.random.default_rng(n * 7 + k)
    t = (n - k) // 2
    for shorten in (0, min(n // 3, k - 1)):
        ks, ns = k - shorten, n - shorten
        M = rng.integers(0, q, (N, ks))
        C = rs.encode(M).numpy().astype(np.int64)
        assert np.array_equal(C, R_.encode(M).astype(np.int64))
        assert np.array_equal(rs.encode(M, output="parity").numpy(), C[:, ks:])
        R = C.copy()
        E = np.zeros((N, ns), dtype=bool)
        for i in range(N):
            ne = int(rng.integers(0, t + 3))
            nu = int(rng.integers(0, n - k + 2)) if i % 3 == 0 else 0
            pos = rng.choice(ns, min(ne, ns), replace=False)
            R[i, pos] = (R[i, pos] + rng.integers(1, q, pos.size)) % q
            if nu:
                epos = rng.choice(ns, min(nu, ns), replace=False)
                E[i, epos] = True
                R[i, epos] = rng.integers(0, q, epos.size)
        dec, nerr = rs.decode(R, erasures=E, output="codeword", errors=True)
        odec, onerr = R_.decode(R, E)
        assert np.array_equal(nerr, onerr), np.nonzero(nerr != onerr)[0][:10]
        assert np.array_equal(dec.numpy().astype(np.int64), odec.astype(np.int64))
        assert dec.dtype == GF.dtypes[0]
        dec2, nerr2 = rs.decode(R, errors=True)
        odec2, onerr2 = R_.decode(R)
        assert np.array_equal(nerr2, onerr2) and np.array_equal(dec2.numpy().astype(np.int64), odec2[:, :ks].astype(np.int64))
        assert np.array_equal(rs.detect(R), R_.detect(R))
    # other storage widths of the same field, and a non-systematic code of the same parameters
    wide = rs.decode(GF(R, dtype=GF.dtypes[-1]), output="codeword")
    assert wide.dtype == GF.dtypes[-1] and np.array_equal(wide.numpy().astype(np.int64), odec2.astype(np.int64))
    if n <= 4095:
        ns_rs = ga.ReedSolomon(n, k, field=GF, c=c, systematic=False)
        M = rng.integers(0, q, (8, k))
        Cn = ns_rs.encode(M)
        assert not ns_rs.detect(Cn).any()
        Rn = Cn.numpy().astype(np.int64)
        Rn[:, 3] = (Rn[:, 3] + 1) % q
        back, ne = ns_rs.decode(Rn, errors=True)
        assert np.array_equal(back.numpy().astype(np.int64), M) and (ne == 1).all()


def test_wide_limits():
    GF = ga.GF(2**10)
    with pytest.raises((NotImplementedError, ValueError)):
        ga.ReedSolomon(1023, 523, field=GF).encode(np.zeros(523, dtype=np.int64))  # d - 1 = 500 roots: beyond the device path
    with pytest.raises((NotImplementedError, ValueError)):
        ga.ReedSolomon(2**21 - 1, 2**21 - 9, field=ga.GF(2**21)).encode(np.zeros(2**21 - 9, dtype=np.int64))


def test_erasure_mask_as_device_tensor():
    import torch

    rs = ga.ReedSolomon(255, 223)
    rng = np.random.default_rng(77)
    C = rs.encode(rng.integers(0, 256, (64, 223))).numpy()
    R = C.copy()
    E = np.zeros(C.shape, dtype=bool)
    for i in range(64):
        epos = rng.choice(255, 20, replace=False)
        E[i, epos] = True
        R[i, epos] = rng.integers(0, 256, 20)
    want = rs.decode(R, erasures=E, output="codeword")
    got = rs.decode(R, erasures=torch.from_numpy(E).cuda(), output="codeword")
    assert np.array_equal(got.numpy(), want.numpy()) and np.array_equal(got.numpy(), C)
    with pytest.raises(TypeError):
        rs.decode(R, erasures=torch.from_numpy(E.astype(np.uint8)).cuda())
    with pytest.raises(ValueError):
        rs.decode(R, erasures=torch.from_numpy(E[:, :10]).cuda())


def test_c_abi_decode_in_place_wide_field():
    """The same aliasing contract on the table-driven path (uint16 symbols over GF(2^10)): the library decodes through a copy."""
    import torch

    from galois_amd import _lib as L

    GF = ga.GF(2**10)
    rs = ga.ReedSolomon(1023, 1003, field=GF)
    rng = np.random.default_rng(78)
    N = 200
    C = rs.encode(rng.integers(0, 1024, (N, 1003))).numpy().astype(np.int64)
    R = C.copy()
    ne = rng.integers(0, 13, N)
    for i in range(N):
        pos = rng.choice(1023, ne[i], replace=False)
        R[i, pos] = (R[i, pos] + rng.integers(1, 1024, ne[i])) % 1024
    F = O.OracleField(2, 10, int(GF.irreducible_poly), GF._primitive_element_int, lookup=True)
    want, wn = O.OracleRS(F, 1023, 1003, alpha=rs.alpha).decode(R)
    buf = torch.from_numpy(R.astype(np.int16)).cuda()
    nerr = torch.empty(N, dtype=torch.int64, device="cuda")
    L.check(L.lib().gfa_rs_decode(rs._handle, buf.data_ptr(), None, 1023, buf.data_ptr(), nerr.data_ptr(), N, L.U16,
                                  torch.cuda.current_stream().cuda_stream), "gfa_rs_decode")
    assert np.array_equal(nerr.cpu().numpy(), wn)
    assert np.array_equal(buf.cpu().numpy().astype(np.int64), want.astype(np.int64))


@pytest.mark.parametrize("wps", ["4", "5", "6", "8"])
def test_wave_decoder_stress_at_every_occupancy_setting(wps):
    """tools/rs_decode_stress.py in its own process per GFA_RS_WPS (the library reads the knob once): 2^20 words per case at
    0 / t / t + 1 errors, with and without erasures, every word checked by property, a sample by the oracle."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rs_decode_stress.py")], cwd=root, capture_output=True, text=True,
                       timeout=1200, env=dict(os.environ, GFA_RS_WPS=wps))
    assert r.returncode == 0 and "rs decode stress ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("poly", [0x11D, 0x11B])
def test_assembled_berlekamp_massey_loop_against_the_compiler_generated_one(poly):
    """bm_run (inline assembly, hand-placed wait states) and a plain loop compiled from the same recurrence run side by side on
    the device over 600 000 syndrome sequences per field (random, LFSR-generated with early termination, zero-ridden; lengths
    1..32 incl. the 32nd-step corner): every lane of both polynomials, gamma, L and the step count must agree."""
    import ctypes
    import torch
    from galois_amd import _lib as L

    GF = ga.GF(2**8, irreducible_poly=poly)
    st = torch.cuda.current_stream().cuda_stream
    for seed in (1, 2, 3):
        bad = ctypes.c_int64(-1)
        L.check(L.lib().gfa_debug_rs_bm_selftest(GF._handle, 200_000, seed, ctypes.byref(bad), st), "gfa_debug_rs_bm_selftest")
        assert bad.value == 0, f"poly {poly:#x} seed {seed}: {bad.value} sequences differ"
