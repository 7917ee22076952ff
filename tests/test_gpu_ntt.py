"""Parity of the device NTT against golden vectors and the oracle.  Bit-exact."""
import json

import numpy as np
import pytest

import galois_amd as ga
from galois_amd import _numtheory as nt
from galois_amd._ntt import fft_batched
from oracle import gf_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_known_answers_from_reference_tests():
    d = H.reference_outputs()
    for p in (5, 13, 17, 769):
        x = [int(v) for v in d[f"ntt/kat{p}/x"]]
        X = ga.ntt(x, modulus=p)
        H.assert_equal_ints(X.numpy(), d[f"ntt/kat{p}/X"], f"kat {p}")
        assert type(X) is ga.GF(p)
        H.assert_equal_ints(ga.intt(X).numpy(), x)
    assert list(ga.ntt([1, 2, 3, 4], modulus=769).numpy()) == [10, 643, 767, 122]  # tests/fields/test_ntt.py:17
    # default modulus (smallest prime m*N+1 > max) and zero padding (tests/fields/test_ntt.py:63-72)
    assert type(ga.ntt([1, 2, 3, 4])) is ga.GF(5)
    H.assert_equal_ints(ga.ntt([1, 2, 3, 4, 5, 6], size=8, modulus=17).numpy(), ga.ntt([1, 2, 3, 4, 5, 6, 0, 0], modulus=17).numpy())
    # unscaled inverse (tests/fields/test_ntt.py:115-135)
    X = ga.ntt([1, 2, 3, 4], modulus=13)
    H.assert_equal_ints(ga.intt(X, scaled=False).numpy(), (np.array([1, 2, 3, 4]) * 4) % 13)


def test_reference_generated_vectors():
    d = H.reference_outputs()
    tags = sorted({k.split("/")[1] for k in d.files if k.startswith("ntt/") and not k.startswith("ntt/kat")})
    for tag in tags:
        order = int(d[f"ntt/{tag}/order"])
        GF = ga.GF(order)
        x = d[f"ntt/{tag}/x"]
        gx = GF(np.array([int(v) for v in x], dtype=object)) if order > 2**63 else GF(x.astype(np.int64))
        for mode in GF.ufunc_modes:
            GF.compile(mode)
            H.assert_equal_ints(np.fft.fft(gx).numpy(), d[f"ntt/{tag}/fft"], tag + mode)
            H.assert_equal_ints(np.fft.ifft(gx).numpy(), d[f"ntt/{tag}/ifft"], tag + mode + " inverse")
        GF.compile("auto")


@pytest.mark.parametrize("order,logn", [(65537, 1), (65537, 5), (65537, 10), (65537, 13), (65537, 14), (65537, 16),
                                        (7340033, 12), (7340033, 17), (7340033, 20), (2**64 - 2**32 + 1, 12),
                                        (2**64 - 2**32 + 1, 13), (2**64 - 2**32 + 1, 16), (2147483647 * 0 + 3221225473, 18),
                                        (18446744069414584321 - 0, 20)])
def test_power_of_two_sizes_against_oracle(order, logn):
    n = 1 << logn
    GF = ga.GF(order)
    F = O.OracleField(order, 1, None, GF._primitive_element_int)
    rng = np.random.default_rng(logn)
    if order < 2**63:
        x = rng.integers(0, order, n, dtype=np.uint64)
        gx = GF(x.astype(GF.dtypes[0] if GF.dtypes != [np.object_] else np.int64))
    else:
        x = (rng.integers(0, 2**63, n, dtype=np.uint64) * np.uint64(2) + np.uint64(1)) % np.uint64(order)
        gx = GF(np.array([int(v) for v in x], dtype=object))
    omega = GF._root_of_unity_int(n)
    want = F.ntt_u32_pow2(x.astype(np.uint32), omega).astype(np.uint64) if order < 2**31 else F.ntt(x, omega=omega)
    X = np.fft.fft(gx)
    H.assert_equal_ints(X.numpy().astype(np.uint64) if order < 2**63 else np.array([int(v) for v in X.numpy()], dtype=np.uint64), want)
    back = np.fft.ifft(X)
    H.assert_equal_ints(np.array([int(v) for v in back.numpy()], dtype=np.uint64), x)


def test_batched_and_24bit_sizes():
    # config C3(i): 16 x 2^16 points over GF(65537)
    GF = ga.GF(65537)
    F = O.OracleField(65537, 1, None, 3)
    x = np.random.default_rng(3).integers(0, 65537, (16, 65536), dtype=np.uint32)
    X = fft_batched(GF(x))
    for i in (0, 7, 15):
        H.assert_equal_ints(X.numpy()[i], F.ntt_u32_pow2(x[i], 3))
    assert np.array_equal(fft_batched(X, inverse=True).numpy(), x)
    # 2^24 points: round trip + linearity (size-independent properties; the oracle would take minutes)
    G = ga.GF(7340033 * 0 + 469762049)  # 7 * 2^26 + 1
    n = 1 << 24
    a = G(np.random.default_rng(1).integers(0, 469762049, n, dtype=np.uint32))
    b = G(np.random.default_rng(2).integers(0, 469762049, n, dtype=np.uint32))
    A, B = np.fft.fft(a), np.fft.fft(b)
    assert np.array_equal(np.fft.ifft(A).numpy(), a.numpy())
    assert np.array_equal(np.fft.fft(a + b).numpy(), (A + B).numpy())
    # X[0] = sum of inputs
    assert int(A[0]) == int(np.add.reduce(a))


@pytest.mark.parametrize("order,n", [(31, 30), (31, 15), (31, 6), (2**8, 255), (2**8, 85), (2**8, 51), (3**5, 242), (3**5, 22),
                                     (5**3, 124), (769, 96), (65537, 3 * 0 + 4096), (7340033, 7 * 64), (2**64 - 2**32 + 1, 3 * 5 * 17),
                                     (127**2, 2304), (7681, 1536), (10753, 1792), (769, 768), (2**12, 4095), (2**10, 341), (3**7, 1093)])
def test_mixed_radix_against_oracle(order, n):
    """tests/fields/test_fft.py:33-104: any length dividing q - 1, any field -- including the sizes of the reference's own FFT
    benchmark (benchmarks/test_fft.py: 256 K points over the first prime-power field with such a root, e.g. 2304 over GF(127^2));
    in lookup mode transforms up to 4096 points over at most 32768 elements run on logarithms (ntt_small_log_kernel)."""
    GF = ga.GF(order)
    p, m = GF.characteristic, GF.degree
    F = O.OracleField(p, m, int(GF.irreducible_poly) if m > 1 else None, GF._primitive_element_int)
    rng = np.random.default_rng(n)
    x = np.array([int(v) % order for v in rng.integers(0, 2**62, n)], dtype=np.uint64)
    gx = GF(np.array([int(v) for v in x], dtype=object)) if order > 2**63 else GF(x.astype(np.int64))
    for mode in GF.ufunc_modes:
        GF.compile(mode)
        X = np.fft.fft(gx)
        H.assert_equal_ints(np.array([int(v) for v in X.numpy()], dtype=np.uint64), F.ntt(x))
        H.assert_equal_ints(np.array([int(v) for v in np.fft.ifft(X).numpy()], dtype=np.uint64), x)
        # FFT == polynomial evaluation at omega^k (tests/fields/test_fft.py:33-45), checked on a few k
        omega = GF._root_of_unity_int(n)
        for k in (0, 1, n - 1):
            pt = F.pow([omega], [k])
            assert int(F.poly_eval(x[::-1].copy(), pt)[0]) == int(X[k])
        # zero padding
        if n >= 6:
            Xp = np.fft.fft(gx[: n // 2], n=n)
            xp = x.copy(); xp[n // 2:] = 0
            H.assert_equal_ints(np.array([int(v) for v in Xp.numpy()], dtype=np.uint64), F.ntt(xp))
    GF.compile("auto")
    with pytest.raises(ValueError):
        np.fft.fft(gx, n=n + 1 if (order - 1) % (n + 1) else 7 * n + 13)
    with pytest.raises(ValueError):
        np.fft.fft(gx, norm="ortho")
    with pytest.raises(ValueError):
        np.fft.fft(gx.reshape(1, -1))


def test_convolve_sage_fixtures_and_ntt_route():
    """np.convolve = polynomial product: Sage fixtures for every field, then long products
    that take the three-NTT route, checked against the oracle's direct convolution."""
    import json
    from tests.test_gpu_elementwise import field_from_props

    for tag in H.SAGE_FIELDS:
        props, d = H.load_sage_field(tag)
        GF = field_from_props(props)
        for mode in GF.ufunc_modes:
            GF.compile(mode)
            for i in range(3):
                z = np.convolve(GF(d[f"convolve{i}_X"].astype(np.int64)), GF(d[f"convolve{i}_Y"].astype(np.int64)))
                assert type(z) is GF
                H.assert_equal_ints(z.numpy(), d[f"convolve{i}_Z"], f"{tag} convolve {mode}")
        GF.compile("auto")
    for order, na, nb in [(7340033, 5000, 3000), (65537, 20000, 9000), (2**64 - 2**32 + 1, 3000, 2500), (31, 3000, 2000)]:
        GF = ga.GF(order)
        F = O.OracleField(order, 1, None, GF._primitive_element_int)
        rng = np.random.default_rng(na)
        a = (rng.integers(0, 2**62, na, dtype=np.uint64) * np.uint64(3)) % np.uint64(order)
        b = (rng.integers(0, 2**62, nb, dtype=np.uint64) * np.uint64(5)) % np.uint64(order)
        mk = (lambda v: GF(np.array([int(t) for t in v], dtype=object))) if order > 2**63 else (lambda v: GF(v.astype(np.int64)))
        z = np.convolve(mk(a), mk(b))
        H.assert_equal_ints(np.array([int(t) for t in z.numpy()], dtype=np.uint64), F.convolve(a, b), f"convolve {order}")
    with pytest.raises(ValueError):
        np.convolve(ga.GF(31)([1, 2]), ga.GF(31)([1, 2]), mode="same")
    with pytest.raises(TypeError):
        np.convolve(ga.GF(31)([1, 2]), ga.GF(7)([1, 2]))


def test_distributed_column_pass_emulated_on_one_gpu():
    """gfa_ntt_columns + batched gfa_ntt composed exactly as galois_amd.dist.ntt_four_step_distributed does, with the
    all-to-all emulated in-process for G ranks on one GPU; result compared with the single-GPU transform."""
    import torch
    from galois_amd import dist as gdist
    from galois_amd import _lib as L

    for order, n1, n2, G in [(7340033, 1 << 10, 1 << 8, 4), (2**64 - 2**32 + 1, 1 << 10, 1 << 6, 8), (65537, 1 << 6, 1 << 10, 2)]:
        GF = ga.GF(order)
        n = n1 * n2
        rng = np.random.default_rng(n1)
        xs = (rng.integers(0, 2**62, n, dtype=np.uint64) * np.uint64(3)) % np.uint64(order)
        native = np.uint64 if order > 2**32 else np.uint32
        omega = GF._root_of_unity_int(n)
        cols, rows = n2 // G, n1 // G
        a_parts = []
        for g in range(G):
            local = torch.from_numpy(gdist.columns_to_local(xs.astype(native), g, G, n1, n2).view(np.int64 if native is np.uint64 else np.int32)).cuda()
            a_parts.append(gdist._device_column_pass(GF, local, n1, cols, g * cols, n, omega))
        omega_n2 = GF._scalar(L.OP_POW, omega, n1)
        outs = []
        for r in range(G):
            mine = torch.cat([a_parts[s][r * rows:(r + 1) * rows, :] for s in range(G)], dim=1).contiguous()
            outs.append(gdist._device_row_pass(GF, mine, n2, omega_n2).cpu().numpy().view(native))
        full = gdist.local_to_natural(outs, n1, n2)
        gx = GF(np.array([int(t) for t in xs], dtype=object)) if order > 2**63 else GF(xs.astype(np.int64))
        want = np.fft.fft(gx).numpy()
        H.assert_equal_ints(full.astype(np.uint64), np.array([int(t) for t in want], dtype=np.uint64), f"dist {order}")


@pytest.mark.parametrize("logn", [5, 10, 15, 20])
def test_unreduced_butterflies_at_the_magnitude_limit(logn):
    """p just below 2^24 (13 * 2^20 + 1) takes the unreduced register butterflies, whose intermediate values reach 96 p:
    worst-case inputs (all p - 1, alternating 0 / p - 1, an impulse) and random data against the oracle."""
    p = 13 * 2**20 + 1
    assert ga.is_prime(p) and p < 2**24
    GF = ga.GF(p)
    F = O.OracleField(p, 1, None, int(GF.primitive_element))
    n = 1 << logn
    omega = GF._root_of_unity_int(n)
    rng = np.random.default_rng(logn)
    rows = [np.full(n, p - 1, dtype=np.uint32), np.tile(np.array([0, p - 1], dtype=np.uint32), n // 2),
            np.concatenate([[p - 1], np.zeros(n - 1, dtype=np.uint32)]).astype(np.uint32),
            rng.integers(0, p, n, dtype=np.uint32), (p - 1 - rng.integers(0, 3, n)).astype(np.uint32)]
    X = ga.fft_batched(GF(np.stack(rows))) if hasattr(ga, "fft_batched") else None
    if X is None:
        from galois_amd._ntt import fft_batched
        X = fft_batched(GF(np.stack(rows)))
    got = X.numpy()
    for i, r in enumerate(rows):
        assert np.array_equal(got[i], F.ntt_u32_pow2(r, omega)), f"row {i}"
    from galois_amd._ntt import fft_batched
    back = fft_batched(X, inverse=True).numpy()
    assert np.array_equal(back, np.stack(rows))


@pytest.mark.parametrize("order,logn", [(469762049, 21), (469762049, 22), (469762049, 23), (2013265921, 21), (2013265921, 24),
                                        (3221225473, 22), (2**64 - 2**32 + 1, 21), (2**64 - 2**32 + 1, 23)])
def test_three_pass_sizes_against_oracle(order, logn):
    """2^21 .. 2^29 points run as three passes of the register kernel (n = L0 * L1 * L2); every output compared."""
    n = 1 << logn
    GF = ga.GF(order)
    F = O.OracleField(order, 1, None, GF._primitive_element_int)
    rng = np.random.default_rng(logn)
    x = rng.integers(0, 2**63, n, dtype=np.uint64) % np.uint64(order)
    gx = GF(x.astype(np.int64)) if order < 2**63 else GF._wrap(__import__("torch").from_numpy(x.view(np.int64)).cuda(), np.object_)
    omega = GF._root_of_unity_int(n)
    want = F.ntt_u32_pow2(x.astype(np.uint32), omega).astype(np.uint64) if order < 2**31 else F.ntt(x, omega=omega)
    X = np.fft.fft(gx)
    got = X._t.cpu().numpy().view(np.uint64) if order > 2**63 else X.numpy().astype(np.uint64)
    assert np.array_equal(got, want)
    back = np.fft.ifft(X)
    gb = back._t.cpu().numpy().view(np.uint64) if order > 2**63 else back.numpy().astype(np.uint64)
    assert np.array_equal(gb, x)


def test_three_pass_batched_and_2_26():
    GF = ga.GF(469762049)
    F = O.OracleField(469762049, 1, None, GF._primitive_element_int)
    n = 1 << 21
    x = np.random.default_rng(5).integers(0, 469762049, (3, n), dtype=np.uint32)
    X = fft_batched(GF(x))
    for i in range(3):
        assert np.array_equal(X.numpy()[i].astype(np.uint32), F.ntt_u32_pow2(x[i], GF._root_of_unity_int(n)))
    # 2^26 points: inverse round trip, X[0] = sum, and a few outputs recomputed as sum_j x_j w^(jk) with the (separately
    # pinned) power, multiply and reduce kernels
    n = 1 << 26
    a = GF(np.random.default_rng(1).integers(0, 469762049, n, dtype=np.uint32))
    A = np.fft.fft(a)
    assert np.array_equal(np.fft.ifft(A).numpy(), a.numpy())
    assert int(A[0]) == int(np.add.reduce(a))
    omega = GF._root_of_unity_int(n)
    j = np.arange(n, dtype=np.int64)
    for k in (1, 2, 12345, (1 << 25) + 17, n - 1):
        wk = GF(omega) ** int(k)
        col = GF(np.full(n, int(wk), dtype=np.int64)) ** j
        assert int(A[k]) == int(np.add.reduce(a * col)), k


@pytest.mark.parametrize("order,na,nb", [(2**31 - 1, 8192, 8192), (4294967291, 8192, 8200), (65521, 70000, 1000), (31, 9000, 9000),
                                         (3, 8192, 8192), (251, 20000, 4000)])
def test_convolve_crt_route_against_oracle(order, na, nb):
    """Long products over prime fields WITHOUT a power-of-two root of unity: gfa_convolve goes through three auxiliary NTT
    primes + CRT (gfa_conv_crt.hip).  Every coefficient against the oracle's O(na*nb) product."""
    GF = ga.GF(order)
    F = O.OracleField(order, 1, None, GF._primitive_element_int)
    rng = np.random.default_rng(na + nb)
    a = rng.integers(0, order, na, dtype=np.uint64)
    b = rng.integers(0, order, nb, dtype=np.uint64)
    # worst case for the CRT bound in one corner: all-(p-1) runs
    a[: na // 4] = order - 1
    b[: nb // 4] = order - 1
    for dt in GF.dtypes[:2]:
        z = np.convolve(GF(a.astype(np.int64), dtype=dt), GF(b.astype(np.int64), dtype=dt))
        assert z.dtype == dt
        H.assert_equal_ints(z.numpy().astype(np.uint64), F.convolve(a, b), f"crt convolve {order} {dt}")


def test_convolve_crt_large_by_evaluation():
    """2^20 x 2^20 terms over GF(2^31 - 1): c(x) = a(x) b(x) at random points (Horner kernel), plus the end coefficients."""
    GF = ga.GF(2**31 - 1)
    rng = np.random.default_rng(11)
    n = 1 << 20
    a = GF(rng.integers(0, 2**31 - 1, n, dtype=np.int64))
    b = GF(rng.integers(0, 2**31 - 1, n - 3, dtype=np.int64))
    c = np.convolve(a, b)
    assert c.size == 2 * n - 4
    assert int(c[0]) == int(a[0] * b[0]) and int(c[-1]) == int(a[-1] * b[-1])
    pts = GF(rng.integers(1, 2**31 - 1, 8, dtype=np.int64))
    # np.convolve's index 0 is the highest-degree coefficient of a Poly (degree-descending), as in Poly.__mul__
    pa, pb, pc = ga.Poly(a), ga.Poly(b), ga.Poly(c)
    assert np.array_equal(pc(pts).numpy(), (pa(pts) * pb(pts)).numpy())


@pytest.mark.parametrize("order,logn", [(3221225473, 28), (2**64 - 2**32 + 1, 28), (3221225473, 29)])
def test_three_pass_largest_sizes(order, logn):
    """The largest single transforms of the three-pass path (tile offsets close to the kernel's 32-bit limit): inverse round
    trip, X[0] = sum, and two outputs recomputed chunk by chunk with the power / multiply / reduce kernels."""
    import torch

    n = 1 << logn
    GF = ga.GF(order)
    g = torch.Generator(device="cuda").manual_seed(logn)
    if order < 2**32:
        t = torch.randint(0, 2**31, (n,), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)  # < 2^31 < p
        a = GF._wrap(t, np.uint32)
    else:
        a = GF._wrap(torch.randint(0, 2**62, (n,), dtype=torch.int64, device="cuda", generator=g), np.object_)
    A = np.fft.fft(a)
    back = np.fft.ifft(A)
    assert bool(torch.equal(back._t, a._t))
    del back
    assert int(A[0]) == int(np.add.reduce(a))
    omega = GF._root_of_unity_int(n)
    chunk = 1 << 24
    j = np.arange(chunk, dtype=np.int64)
    for k in (1, n - 3):
        wk = GF(omega) ** int(k)
        col = GF(np.full(chunk, int(wk), dtype=np.int64 if order < 2**63 else object)) ** j
        step = wk ** chunk
        acc, scale = GF(0), GF(1)
        for c in range(n // chunk):
            part = np.add.reduce(a[c * chunk:(c + 1) * chunk] * col)
            acc = acc + part * scale
            scale = scale * step
        assert int(A[k]) == int(acc), (order, logn, k)


def test_distributed_transform_over_rccl_world_of_one():
    """galois_amd.dist.ntt_four_step_distributed itself -- device kernels AND the RCCL all_to_all_single on device tensors --
    in a process group of one rank (the box has one GPU; the N > 1 exchange layout is covered by the gloo tests)."""
    import os
    import torch
    import torch.distributed as tdist
    from galois_amd import dist as gdist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    tdist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for order, n1, n2 in [(2**64 - 2**32 + 1, 1 << 10, 1 << 12), (7340033, 1 << 8, 1 << 10)]:
            GF = ga.GF(order)
            n = n1 * n2
            xs = (np.random.default_rng(n2).integers(0, 2**62, n, dtype=np.uint64) * np.uint64(5)) % np.uint64(order)
            native = np.uint64 if order > 2**32 else np.uint32
            local = torch.from_numpy(gdist.columns_to_local(xs.astype(native), 0, 1, n1, n2).view(np.int64 if native is np.uint64 else np.int32)).cuda()
            out = gdist.ntt_four_step_distributed(GF, local, n1, n2)
            full = gdist.local_to_natural([out.cpu().numpy().view(native)], n1, n2)
            gx = GF._wrap(torch.from_numpy(xs.view(np.int64)).cuda(), np.object_) if order > 2**63 else GF(xs.astype(np.int64))
            want = np.fft.fft(gx)
            wv = want._t.cpu().numpy().view(np.uint64) if order > 2**63 else want.numpy().astype(np.uint64)
            assert np.array_equal(full.astype(np.uint64), wv), order
    finally:
        tdist.destroy_process_group()


def test_fermat_single_pass_kernel_gf65537():
    """2^16-point transforms over GF(65537) in batches >= 64 run on the one-pass register kernel (gfa_ntt_fermat.hip):
    random and worst-case inputs, several roots of unity (every odd power of w is another primitive root, which
    exercises the input-order permutation), forward and scaled inverse, against the oracle and the two-pass kernel."""
    import torch
    from galois_amd import _lib as L

    lib = L.lib()
    GF = ga.GF(65537)
    F = O.OracleField(65537, 1, None, 3)
    n, batch = 65536, 64
    rng = np.random.default_rng(16)
    x = rng.integers(0, 65537, (batch, n), dtype=np.uint32)
    x[0] = 65536                      # every element -1
    x[1] = 0
    x[2, ::2] = 65536; x[2, 1::2] = 0
    x[3, ::2] = 0; x[3, 1::2] = 65536
    x[4] = rng.choice(np.array([0, 1, 65535, 65536], dtype=np.uint32), n)
    x[5, : n // 2] = 65536; x[5, n // 2:] = 1
    xt = torch.from_numpy(x.view(np.int32)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    w = GF._root_of_unity_int(n)
    for j in (1, 3, 65535, 12345, 40961):
        wj = pow(w, j, 65537)
        out = torch.empty_like(xt)
        L.check(lib.gfa_ntt(GF._handle, xt.data_ptr(), out.data_ptr(), n, batch, wj, 0, L.U32, st))
        got = out.cpu().numpy().view(np.uint32)
        for i in (0, 1, 2, 3, 4, 5, 17, 63):
            H.assert_equal_ints(got[i], F.ntt_u32_pow2(x[i], wj), f"root w^{j}, row {i}")
        # every row against the two-pass kernel (batches below 64 take that path)
        ref = torch.empty_like(xt)
        for b0 in range(0, batch, 32):
            L.check(lib.gfa_ntt(GF._handle, xt[b0:b0 + 32].data_ptr(), ref[b0:b0 + 32].data_ptr(), n, 32, wj, 0, L.U32, st))
        assert torch.equal(out, ref), f"root w^{j}"
        # scaled inverse with the inverse root, in place
        winv = pow(wj, 65537 - 2, 65537)
        L.check(lib.gfa_ntt(GF._handle, out.data_ptr(), out.data_ptr(), n, batch, winv, 1, L.U32, st))
        assert torch.equal(out, xt), f"inverse, root w^{j}"
    # through the array front end
    X = fft_batched(GF(x))
    H.assert_equal_ints(X.numpy()[9], F.ntt_u32_pow2(x[9], w))
    assert np.array_equal(fft_batched(X, inverse=True).numpy(), x)


@pytest.mark.parametrize("logn", [10, 11, 12, 13, 14, 15])
def test_fermat_grouped_kernel_gf65537_2e10_to_2e15(logn):
    """r06: transforms of 2^10 .. 2^15 points over GF(65537) with at least 2^22 points in the batch run G = 2^16 / n to a workgroup on
    the one-pass kernel (gfa_ntt_fermat.hip, LOGG > 0: G first networks of radix 64 / G, everything after them shared with the 2^16-point
    form).  Worst-case rows and random ones, several roots (every odd power: the input-order permutation), batches that are NOT a
    multiple of the group (the last block's missing transforms read zeros and store nothing -- the words after the batch must stay
    untouched), forward against the oracle and the small-batch route, the scaled inverse (1 / n = -2^(16 - log n)) in place."""
    import torch
    from galois_amd import _lib as L

    lib = L.lib()
    p = 65537
    GF = ga.GF(p)
    F = O.OracleField(p, 1, None, 3)
    n = 1 << logn
    G = 65536 // n
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(logn)
    w = GF._root_of_unity_int(n)
    for batch, j in ((64 * G, 1), (64 * G + 1, 3), (65 * G + G - 1, n - 1), (130 * G + 3, 12345)):
        wj = pow(w, j | 1, p)
        x = rng.integers(0, p, (batch, n), dtype=np.uint32)
        x[0] = 65536
        x[1, ::2] = 65536; x[1, 1::2] = 0
        x[batch - 1] = rng.choice(np.array([0, 1, 65535, 65536], dtype=np.uint32), n)
        xt = torch.from_numpy(x.view(np.int32)).cuda()
        buf = torch.full((batch * n + 4096,), -7, dtype=torch.int32, device="cuda")  # guard words behind the batch
        out = buf[:batch * n].view(batch, n)
        L.check(lib.gfa_ntt(GF._handle, xt.data_ptr(), out.data_ptr(), n, batch, wj, 0, L.U32, st))
        assert bool((buf[batch * n:] == -7).all()), "stores behind the last transform"
        got = out.cpu().numpy().view(np.uint32)
        for i in {0, 1, G - 1, G, batch // 2, batch - 2, batch - 1}:
            H.assert_equal_ints(got[i], F.ntt_u32_pow2(x[i], wj), f"2^{logn}, batch {batch}, row {i}")
        ref = torch.empty_like(xt)  # every row against the route small batches take
        step = max(1, (1 << 21) // n)
        for b0 in range(0, batch, step):
            b1 = min(batch, b0 + step)
            L.check(lib.gfa_ntt(GF._handle, xt[b0:b1].data_ptr(), ref[b0:b1].data_ptr(), n, b1 - b0, wj, 0, L.U32, st))
        assert torch.equal(out, ref), f"2^{logn}, batch {batch}"
        L.check(lib.gfa_ntt(GF._handle, out.data_ptr(), out.data_ptr(), n, batch, pow(wj, p - 2, p), 1, L.U32, st))
        assert torch.equal(out, xt), f"scaled inverse, 2^{logn}, batch {batch}"
        assert bool((buf[batch * n:] == -7).all())


def _emulate_forward(GF, xs_dev_cols, n1, n2, G, omega):
    """G ranks' work of galois_amd.dist.ntt_four_step_distributed on one GPU (the all-to-all is a re-slicing)."""
    from galois_amd import dist as gdist
    from galois_amd import _lib as L
    import torch

    cols, rows = n2 // G, n1 // G
    a_parts = [gdist._device_column_pass(GF, xs_dev_cols[g], n1, cols, g * cols, n1 * n2, omega) for g in range(G)]
    omega_n2 = GF._scalar(L.OP_POW, omega, n1)
    outs = []
    for r in range(G):
        # what the all-to-all delivers to rank r: recv[s][k1_local][c]; the row pass reads those per-peer chunks in place
        recv = torch.stack([a_parts[s][r * rows:(r + 1) * rows, :] for s in range(G)]).contiguous()
        got = gdist._device_row_pass_from_chunks(GF, recv, n2, omega_n2)
        mine = recv.permute(1, 0, 2).reshape(rows, n2).contiguous()
        ref = gdist._device_row_pass(GF, mine, n2, omega_n2)
        if got is not None:
            assert torch.equal(got, ref), "chunked row pass differs from the row pass on the re-laid-out copy"
        outs.append(ref)
    return outs


def _emulate_inverse(GF, row_blocks, n1, n2, G, omega):
    """G ranks' work of galois_amd.dist.intt_four_step_distributed on one GPU."""
    from galois_amd import dist as gdist
    from galois_amd import _lib as L
    import torch

    cols, rows = n2 // G, n1 // G
    omega_inv = GF._scalar(L.OP_RECIP, omega, 0)
    w_rows = GF._scalar(L.OP_POW, omega_inv, n1)
    b_parts = [gdist._device_row_pass(GF, row_blocks[g], n2, w_rows) for g in range(G)]
    for g in range(G):
        # the row pass that writes the send buffer send[s][k1_local][c] directly
        send = gdist._device_row_pass_to_chunks(GF, row_blocks[g], n2, w_rows, G)
        if send is not None:
            assert torch.equal(send, b_parts[g].view(rows, G, cols).permute(1, 0, 2).contiguous()), "chunked send buffer differs"
    outs = []
    for s in range(G):
        mine = torch.cat([b_parts[g][:, s * cols:(s + 1) * cols] for g in range(G)], dim=0).contiguous()
        outs.append(gdist._device_column_pass_inv(GF, mine, n1, cols, s * cols, n1 * n2, omega_inv, True))
    return outs


def test_distributed_inverse_emulated_on_one_gpu():
    """intt_four_step_distributed's kernels (batched gfa_ntt + gfa_ntt_columns_inv) consume the forward transform's row-block
    layout and return its column-block input bit for bit, for 32-bit (lazy / unreduced Shoup) and 64-bit fields."""
    import torch
    from galois_amd import dist as gdist

    for order, n1, n2, G in [(7340033, 1 << 10, 1 << 8, 4), (2**64 - 2**32 + 1, 1 << 10, 1 << 6, 8), (65537, 1 << 6, 1 << 10, 2),
                             (469762049, 1 << 9, 1 << 11, 2), (3221225473, 1 << 8, 1 << 6, 4)]:
        GF = ga.GF(order)
        n = n1 * n2
        rng = np.random.default_rng(n2)
        xs = (rng.integers(0, 2**62, n, dtype=np.uint64) * np.uint64(3)) % np.uint64(order)
        xs[:4] = order - 1
        native = np.uint64 if order > 2**32 else np.uint32
        tview = np.int64 if native is np.uint64 else np.int32
        omega = GF._root_of_unity_int(n)
        locals_ = [torch.from_numpy(gdist.columns_to_local(xs.astype(native), g, G, n1, n2).view(tview)).cuda() for g in range(G)]
        fwd = _emulate_forward(GF, locals_, n1, n2, G, omega)
        back = _emulate_inverse(GF, fwd, n1, n2, G, omega)
        for g in range(G):
            assert torch.equal(back[g], locals_[g]), f"order {order}, rank {g}"
    # the chunked layouts are really taken by the kernel at the C5 shape (not silently refused)
    GL = ga.GF(2**64 - 2**32 + 1)
    recv = torch.zeros((8, 2, 1 << 13), dtype=torch.int64, device="cuda")
    assert gdist._device_row_pass_from_chunks(GL, recv, 1 << 16, GL._root_of_unity_int(1 << 16)) is not None
    assert gdist._device_row_pass_to_chunks(GL, torch.zeros((2, 1 << 16), dtype=torch.int64, device="cuda"), 1 << 16,
                                            GL._root_of_unity_int(1 << 16), 8) is not None


def test_c5_full_size_emulated_on_one_gpu():
    """BASELINE config C5 at its real size: 2^26 Goldilocks points split 1024 x 65536 over G = 8 ranks, every rank's kernels
    run on this GPU with the exchange emulated.  The ORACLE (oracle/gf_oracle.c, the reference's fft_jit restated) transforms
    the same 2^26 points on the host -- about a minute -- and EVERY output of both the single-GPU three-pass transform and
    the four-step emulation is compared with it; then the distributed inverse takes the result back."""
    import torch
    from galois_amd import dist as gdist

    order = 2**64 - 2**32 + 1
    GF = ga.GF(order)
    n1, n2, G = 1 << 10, 1 << 16, 8
    assert gdist.choose_split(1 << 26, G) == (n1, n2)
    n = n1 * n2
    x = torch.empty(n, dtype=torch.int64, device="cuda").random_(0, 2**62) * 3
    # reduce into the field on the device: (x mod p) via the field's own add of zero is not needed -- values < 2^64 - 2^32 + 1
    x = torch.where(x < 0, x + (2**32 - 1), x)  # bit patterns >= 2^63 are valid uint64 values; keep them below p
    xu = x.cpu().numpy().view(np.uint64)
    xu %= np.uint64(order)
    x = torch.from_numpy(xu.view(np.int64)).cuda()
    omega = GF._root_of_unity_int(n)
    gx = GF._wrap(x, np.object_)
    want = np.fft.fft(gx)._t  # single-GPU path
    F = O.OracleField(order, 1, None, int(GF.primitive_element))
    oracle_X = F.ntt(xu.copy(), omega=omega)
    assert np.array_equal(want.cpu().numpy().view(np.uint64), oracle_X), "single-GPU 2^26-point transform differs from the oracle"
    del oracle_X  # (from here on `want` IS the oracle's vector)
    cols = n2 // G
    xm = x.view(n1, n2)
    locals_ = [xm[:, g * cols:(g + 1) * cols].contiguous() for g in range(G)]
    fwd = _emulate_forward(GF, locals_, n1, n2, G, omega)
    rows = n1 // G
    wv = want.view(n2, n1)  # X[k1 + n1*k2] -> [k2][k1]
    for g in range(G):
        assert torch.equal(fwd[g], wv[:, g * rows:(g + 1) * rows].t().contiguous()), f"rank {g}"
    # size-independent properties of the single-GPU reference itself: X[0] = sum x, X[N/2] = alternating sum
    s = sum(int(v) for v in np.add.reduce(xu.reshape(-1, 1 << 10).astype(object), axis=1)) % order
    assert int(want.cpu().numpy().view(np.uint64)[0]) == s
    back = _emulate_inverse(GF, fwd, n1, n2, G, omega)
    for g in range(G):
        assert torch.equal(back[g], locals_[g]), f"inverse, rank {g}"


def test_c_abi_distributed_transform_owns_its_rccl_exchange():
    """gfa_ntt_dist / gfa_intt_dist (the collective lives inside the library: columns, ONE ncclAllToAll, chunked rows) driven
    with a communicator created directly through RCCL, as a non-Python host would; one rank here (the exchange is then a
    device copy inside RCCL), compared with the single-GPU transform and inverted back."""
    import ctypes
    import os
    import torch
    from galois_amd import _lib as L

    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    rccl = ctypes.CDLL(path if os.path.exists(path) else "librccl.so", mode=ctypes.RTLD_GLOBAL)

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        lib = L.lib()
        st = torch.cuda.current_stream().cuda_stream
        for order, n1, n2, dt, tdt in [(2**64 - 2**32 + 1, 1 << 10, 1 << 12, L.U64, torch.int64), (7340033, 1 << 8, 1 << 10, L.U32, torch.int32),
                                       (469762049, 1 << 10, 1 << 16, L.U32, torch.int32),
                                       # rows of 2^21 points: the in-place chunked row kernel declines (n2 > 2^20) and the C path
                                       # re-lays the chunks out and runs the plain transform (the fallback _dist.py has)
                                       (469762049, 4, 1 << 21, L.U32, torch.int32)]:
            GF = ga.GF(order)
            n = n1 * n2
            x = torch.empty(n, dtype=torch.int64, device="cuda").random_(0, min(order, 2**62)).to(tdt)
            omega = GF._root_of_unity_int(n)
            out = torch.empty_like(x)
            L.check(lib.gfa_ntt_dist(GF._handle, comm, 0, 1, x.data_ptr(), out.data_ptr(), n1, n2, omega, dt, st), "gfa_ntt_dist")
            want = torch.empty_like(x)
            L.check(lib.gfa_ntt(GF._handle, x.data_ptr(), want.data_ptr(), n, 1, omega, 0, dt, st))
            assert torch.equal(out.view(n1, n2), want.view(n2, n1).t()), f"forward, order {order}"
            back = torch.empty_like(x)
            L.check(lib.gfa_intt_dist(GF._handle, comm, 0, 1, out.data_ptr(), back.data_ptr(), n1, n2, omega, 1, dt, st), "gfa_intt_dist")
            assert torch.equal(back, x), f"inverse, order {order}"
        torch.cuda.synchronize()
    finally:
        rccl.ncclCommDestroy(comm)


@pytest.mark.parametrize("logn", [6, 10, 13, 20])
def test_goldilocks_lazy_arithmetic_at_the_extremes(logn):
    """The 96-bit lazy butterflies of the Goldilocks kernel on worst-case inputs (all p - 1, alternating 0 / p - 1, words that
    sit on the 2^32 limb boundaries, an impulse) and random data, every output against the oracle; forward and inverse."""
    p = 2**64 - 2**32 + 1
    GF = ga.GF(p)
    F = O.OracleField(p, 1, None, GF._primitive_element_int)
    n = 1 << logn
    omega = GF._root_of_unity_int(n)
    rng = np.random.default_rng(logn)
    edge = np.array([p - 1, p - 2, 2**32 - 1, 2**32, 2**32 + 1, 2**64 - 2**32, 0, 1, 2**63, 0xFFFFFFFF00000000 - 1], dtype=np.uint64)
    rows = [np.full(n, p - 1, dtype=np.uint64), np.tile(np.array([0, p - 1], dtype=np.uint64), n // 2),
            np.concatenate([[p - 1], np.zeros(n - 1)]).astype(np.uint64), rng.choice(edge, n),
            (rng.integers(0, 2**63, n, dtype=np.uint64) * 2 + 1) % np.uint64(p)]
    import torch
    from galois_amd._ntt import fft_batched

    X = fft_batched(GF._wrap(torch.from_numpy(np.stack(rows).view(np.int64)).cuda(), np.object_))
    got = X._t.cpu().numpy().view(np.uint64)
    for i, r in enumerate(rows):
        assert np.array_equal(got[i], F.ntt(r.copy(), omega=omega)), f"row {i}"
    back = fft_batched(X, inverse=True)._t.cpu().numpy().view(np.uint64)
    assert np.array_equal(back, np.stack(rows))


@pytest.mark.parametrize("p,logn", [(28311553, l) for l in range(5, 21)] + [(67043329, l) for l in (5, 9, 10, 11, 13, 16)]
                         + [(132120577, l) for l in range(5, 21)] + [(257949697, l) for l in range(5, 21)] + [(531628033, l) for l in range(5, 21)]
                         + [(q, l) for q in (268369921, 536608769) for l in (5, 8, 10, 11, 14, 16)])
def test_signed_montgomery_kernel_every_line_shape_at_the_magnitude_limit(p, logn):
    """gfa_ntt_m32.hip (odd p < 2^29): int32 representatives that are never reduced inside a radix-32 network reach 32 p.
    28311553 = 27 * 2^20 + 1 covers every line shape up to 2^20 points, 67043329 = 1023 * 2^16 + 1 sits 0.1 % below 2^26.
    r05 -- one prime per bit length 27, 28, 29 through every line shape (132120577 = 126 * 2^20 + 1, 257949697 = 246 * 2^20 + 1:
    the BMAX = 8 schedule of gfa_m32_net.h; 531628033 = 507 * 2^20 + 1: BMAX = 4), and the two primes that sit 0.02-0.05 % below
    2^28 and 2^29 (268369921 = 4095 * 2^16 + 1, 536608769 = 8188 * 2^16 + 1) at the shapes their 2-adicity allows.
    Worst-case rows (all p - 1, alternating 0 / p - 1, an impulse, values within 3 of p) plus random rows, five rows so that
    the single-pass form runs a partly filled tile; forward against the oracle, inverse (1/n folded into the last product)
    as a round trip."""
    assert ga.is_prime(p) and p < 2**29 and (p - 1) % (1 << logn) == 0
    GF = ga.GF(p)
    F = O.OracleField(p, 1, None, int(GF.primitive_element))
    n = 1 << logn
    omega = GF._root_of_unity_int(n)
    rng = np.random.default_rng(logn)
    rows = [np.full(n, p - 1, dtype=np.uint32), np.tile(np.array([0, p - 1], dtype=np.uint32), n // 2),
            np.concatenate([[p - 1], np.zeros(n - 1, dtype=np.uint32)]).astype(np.uint32),
            rng.integers(0, p, n, dtype=np.uint32), (p - 1 - rng.integers(0, 3, n)).astype(np.uint32)]
    X = fft_batched(GF(np.stack(rows)))
    got = X.numpy()
    for i, r in enumerate(rows):
        assert np.array_equal(got[i], F.ntt_u32_pow2(r, omega)), f"row {i}"
    assert np.array_equal(fft_batched(X, inverse=True).numpy(), np.stack(rows))
    # the 1-D front end (one row, np.fft.fft / ifft) takes the same kernels
    one = GF(rows[3])
    assert np.array_equal(np.fft.fft(one).numpy(), got[3])
    assert np.array_equal(np.fft.ifft(np.fft.fft(one)).numpy(), rows[3])


@pytest.mark.parametrize("p", [7340033, 33292289, 67043329, 268369921, 469762049, 536608769])
def test_2e16_points_in_one_workgroup_over_generic_primes(p):
    """2^16-point transforms over odd p < 2^29 in batches >= 64 run as ONE pass over HBM (ntt_m32_2e16_kernel: 64 points per
    thread, radix 64 x 32 x 32).  33292289 = 508 * 2^16 + 1 sits 0.8 % below 2^25, the bound up to which the radix-64 network
    needs no reduction; above it (67043329, 268369921: the BMAX = 8 schedule; 469762049, 536608769: BMAX = 4) operands are
    brought back inside the networks where gfa_m32_net.h's bookkeeping says so.  Worst-case rows (all p - 1,
    alternating 0 / p - 1, an impulse, values within 3 of p) against the oracle, every row against the two-pass kernels
    (batches below 64 take those), scaled inverse in place as a round trip, 65 rows so that a persistent workgroup runs a
    second, shorter round."""
    import torch
    from galois_amd import _lib as L

    lib = L.lib()
    GF = ga.GF(p)
    F = O.OracleField(p, 1, None, int(GF.primitive_element))
    n, batch = 1 << 16, 65
    rng = np.random.default_rng(p & 0xffff)
    x = rng.integers(0, p, (batch, n), dtype=np.uint32)
    x[0] = p - 1
    x[1, ::2] = 0; x[1, 1::2] = p - 1
    x[2] = 0; x[2, 0] = p - 1
    x[3] = (p - 1 - rng.integers(0, 3, n)).astype(np.uint32)
    x[4, : n // 2] = p - 1; x[4, n // 2:] = 1
    xt = torch.from_numpy(x.view(np.int32)).cuda()
    st = torch.cuda.current_stream().cuda_stream
    w = GF._root_of_unity_int(n)
    for j in (1, 3):
        wj = pow(w, j, p)
        out = torch.empty_like(xt)
        L.check(lib.gfa_ntt(GF._handle, xt.data_ptr(), out.data_ptr(), n, batch, wj, 0, L.U32, st))
        got = out.cpu().numpy().view(np.uint32)
        for i in (0, 1, 2, 3, 4, 17, 64):
            H.assert_equal_ints(got[i], F.ntt_u32_pow2(x[i], wj), f"p={p} root w^{j}, row {i}")
        ref = torch.empty_like(xt)
        for b0 in range(0, batch, 13):
            b1 = min(batch, b0 + 13)
            L.check(lib.gfa_ntt(GF._handle, xt[b0:b1].data_ptr(), ref[b0:b1].data_ptr(), n, b1 - b0, wj, 0, L.U32, st))
        assert torch.equal(out, ref), f"p={p} root w^{j}: one-pass and two-pass kernels differ"
        L.check(lib.gfa_ntt(GF._handle, out.data_ptr(), out.data_ptr(), n, batch, pow(wj, p - 2, p), 1, L.U32, st))
        assert torch.equal(out, xt), f"p={p} inverse, root w^{j}"


@pytest.mark.parametrize("p,logn", [(23068673, 21), (132120577, 21), (257949697, 21), (415236097, 22), (377487361, 23), (167772161, 24)])
def test_three_pass_signed_montgomery_transforms_against_the_oracle(p, logn):
    """2^21 .. 2^28 points over odd p < 2^29 (r05): three passes of ntt_m32_kernel -- the first with a SPLIT progression table
    (2^14 .. 2^19 columns), the second in place per row, the third stored transposed.  One prime per class (p < 2^26; BMAX 8; BMAX 4),
    every output against oracle/gf_oracle.c, the scaled inverse as a round trip, and a batch of two (transforms run one after the
    other through one work buffer).  2^26 points over GF(469762049): the next test."""
    import torch

    assert ga.is_prime(p) and (p - 1) % (1 << logn) == 0
    GF = ga.GF(p)
    F = O.OracleField(p, 1, None, int(GF.primitive_element))
    n = 1 << logn
    omega = GF._root_of_unity_int(n)
    rng = np.random.default_rng(logn)
    x = rng.integers(0, p, n, dtype=np.uint32)
    x[:6] = (p - 1, 0, p - 1, 1, p - 2, p - 1)
    want = F.ntt_u32_pow2(x, omega)
    X = np.fft.fft(GF(x))
    assert np.array_equal(X.numpy(), want)
    assert np.array_equal(np.fft.ifft(X).numpy(), x)
    if logn <= 22:
        worst = np.full(n, p - 1, dtype=np.uint32)
        two = fft_batched(GF(np.stack([worst, x])))
        assert np.array_equal(two.numpy()[1], want)
        assert np.array_equal(two.numpy()[0], F.ntt_u32_pow2(worst, omega))
        assert np.array_equal(fft_batched(two, inverse=True).numpy(), np.stack([worst, x]))


def test_2e26_points_over_a_32_bit_prime_against_the_oracle_in_full():
    """The other half of the C5 pin: 2^26 points over GF(469762049) (7 * 2^26 + 1, the CRT-convolution prime and a default
    `galois.ntt` modulus for large inputs, _ntt.py:250-254), single-GPU transform and the four-step emulation over 8 ranks,
    every output against oracle/gf_oracle.c."""
    import torch
    from galois_amd import dist as gdist

    p = 469762049
    GF = ga.GF(p)
    F = O.OracleField(p, 1, None, int(GF.primitive_element))
    G = 8
    n = 1 << 26
    n1, n2 = gdist.choose_split(n, G)
    rng = np.random.default_rng(26)
    xu = rng.integers(0, p, n, dtype=np.uint32)
    xu[:4] = (p - 1, 0, p - 1, 1)
    omega = GF._root_of_unity_int(n)
    want = F.ntt_u32_pow2(xu, omega)
    x = torch.from_numpy(xu.view(np.int32)).cuda()
    got = np.fft.fft(GF._wrap(x, np.uint32))._t
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want), "single-GPU 2^26-point transform differs from the oracle"
    cols, rows = n2 // G, n1 // G
    xm = x.view(n1, n2)
    locals_ = [xm[:, g * cols:(g + 1) * cols].contiguous() for g in range(G)]
    fwd = _emulate_forward(GF, locals_, n1, n2, G, omega)
    wv = torch.from_numpy(want.view(np.int32)).cuda().view(n2, n1)  # X[k1 + n1*k2] -> [k2][k1]
    for g in range(G):
        assert torch.equal(fwd[g], wv[:, g * rows:(g + 1) * rows].t().contiguous()), f"rank {g}"
