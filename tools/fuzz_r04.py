"""Differential fuzzing of the round-4 kernels against the oracle.  Usage: python tools/fuzz_r04.py [seconds] [seed]
  * ntt_m32_2e16_kernel: batches of 64..200 transforms of 2^16 points over random primes c * 2^16 + 1 < 2^25 with random primitive
    roots: a few rows against the oracle, every row against the two-pass kernels (the same batch in pieces below 64 transforms),
    scaled inverse as a round trip;
  * reciprocal / division of random 32-bit primes (skewed-Montgomery batch inversion), zeros among the operands, lengths around the
    vector and batch boundaries, scalar operands;
  * products / quotients of fields with 32768 < q <= 65536 at >= 2^22 elements (the two streaming passes);
  * RS(255,223) and binary BCH(255,223) encode of >= 2^18 words (rs_lfsr_reg_kernel), parity-only and full, against the oracle on a sample
    and against the staged kernel (the same words in pieces below 2^18)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
from oracle import gf_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(seed)
lib = L.lib(); st = torch.cuda.current_stream().cuda_stream
P16 = [c * 65536 + 1 for c in range(1, 512) if ga.is_prime(c * 65536 + 1)]
t_end = time.time() + budget
counts = {"ntt16": 0, "recip": 0, "big16": 0, "rs": 0}
rs = ga.ReedSolomon(255, 223); bch = ga.BCH(255, 223)
F8 = O.OracleField(2, 8, 285, 2, lookup=True); ORS = O.OracleRS(F8, 255, 223); OBCH = O.OracleBCH(F8, 255, 223)
while time.time() < t_end:
    kind = rng.integers(0, 4)
    if kind == 0:
        p = int(rng.choice(P16)); GF = ga.GF(p); F = O.OracleField(p, 1, None, int(GF.primitive_element))
        n = 1 << 16; batch = int(rng.integers(64, 200))
        w = pow(GF._root_of_unity_int(n), int(rng.integers(0, n // 2)) * 2 + 1, p)
        x = rng.integers(0, p, (batch, n), dtype=np.uint32)
        x[int(rng.integers(0, batch))] = p - 1
        xt = torch.from_numpy(x.view(np.int32)).cuda(); out = torch.empty_like(xt); ref = torch.empty_like(xt)
        L.check(lib.gfa_ntt(GF._handle, xt.data_ptr(), out.data_ptr(), n, batch, w, 0, L.U32, st))
        step = int(rng.integers(1, 64))
        for b0 in range(0, batch, step):
            b1 = min(batch, b0 + step)
            L.check(lib.gfa_ntt(GF._handle, xt[b0:b1].data_ptr(), ref[b0:b1].data_ptr(), n, b1 - b0, w, 0, L.U32, st))
        assert torch.equal(out, ref), ("ntt16 one-pass vs two-pass", p, batch)
        got = out.cpu().numpy().view(np.uint32)
        for i in rng.choice(batch, 2, replace=False):
            assert np.array_equal(got[i], F.ntt_u32_pow2(x[i], w)), ("ntt16 vs oracle", p, int(i))
        L.check(lib.gfa_ntt(GF._handle, out.data_ptr(), out.data_ptr(), n, batch, pow(w, p - 2, p), 1, L.U32, st))
        assert torch.equal(out, xt), ("ntt16 inverse", p)
        counts["ntt16"] += 1
    elif kind == 1:
        while True:
            p = int(rng.integers(3, 2**32)) | 1
            if ga.is_prime(p):
                break
        GF = ga.GF(p); F = O.OracleField(p, 1, None, int(GF.primitive_element))
        n = int(rng.choice([1, 3, 4, 15, 16, 17, 63, 64, 65, 1023, 4096, 4099, 70001, 262147]))
        a = rng.integers(0, p, n, dtype=np.uint64); b = rng.integers(0, p, n, dtype=np.uint64)
        bnz = np.where(b == 0, np.uint64(1), b)
        A, B = GF(a.astype(np.uint32)), GF(bnz.astype(np.uint32))
        u = lambda v: v.numpy().astype(np.uint64)
        assert np.array_equal(u(np.reciprocal(B)), F.recip(bnz)), ("recip", p, n)
        assert np.array_equal(u(A / B), F.div(a, bnz)), ("div", p, n)
        assert np.array_equal(u(A / B[n // 2]), F.div(a, np.full(n, bnz[n // 2], dtype=np.uint64))), ("div scalar", p, n)
        if n > 2:
            bz = b.copy(); bz[n // 3] = 0
            try:
                np.reciprocal(GF(bz.astype(np.uint32)))
                raise AssertionError(("no ZeroDivisionError", p, n))
            except ZeroDivisionError:
                pass
        counts["recip"] += 1
    elif kind == 2:
        q = int(rng.choice([2**16, 3**10, 65521, 251**2, 241**2]))
        GF = ga.GF(q); GF.compile("jit-lookup" if rng.random() < 0.7 else "auto")
        F = O.OracleField(GF.characteristic, GF.degree, int(GF.irreducible_poly) if GF.degree > 1 else None, int(GF.primitive_element), lookup=True)
        n = (1 << 22) + int(rng.integers(0, 100_000))
        a = rng.integers(0, q, n, dtype=np.uint64); b = rng.integers(0, q, n, dtype=np.uint64)
        a[rng.integers(0, n, 50)] = 0
        bnz = np.where(b == 0, np.uint64(1), b)
        A, B, Bnz = GF(a.astype(np.uint16)), GF(b.astype(np.uint16)), GF(bnz.astype(np.uint16))
        u = lambda v: v.numpy().astype(np.uint64)
        assert np.array_equal(u(A * B), F.mul(a, b)), ("big16 mul", q, n)
        assert np.array_equal(u(A / Bnz), F.div(a, bnz)), ("big16 div", q, n)
        GF.compile("auto")
        counts["big16"] += 1
    else:
        code, orc, sym = (rs, ORS, 256) if rng.random() < 0.6 else (bch, OBCH, 2)
        B = (1 << 18) + int(rng.integers(0, 3000))
        M = rng.integers(0, sym, (B, 223), dtype=np.uint8)
        C = code.encode(M).numpy()
        Pp = code.encode(M, output="parity").numpy()
        assert np.array_equal(C[:, :223], M) and np.array_equal(C[:, 223:], Pp), "rs parity / full"
        half = code.encode(M[: 1 << 17]).numpy()  # below 2^18 words: the staged kernel
        assert np.array_equal(half, C[: 1 << 17]), "rs reg vs staged"
        sel = rng.choice(B, 256, replace=False)
        want = orc.encode_u8(M[sel]) if sym == 256 else orc.encode(M[sel])
        assert np.array_equal(np.asarray(want, dtype=np.uint8), C[sel]), "rs vs oracle"
        counts["rs"] += 1
print(f"fuzz_r04: {counts['ntt16']} batches of 2^16-point transforms, {counts['recip']} prime-field reciprocal / division cases, {counts['big16']} big-table "
      f"product / quotient arrays, {counts['rs']} large encode batches: every result identical to the oracle (seed {seed}, {budget:.0f} s)")
