"""Differential fuzzing of (a) gfa_convolve's three-prime CRT route over random prime fields below 2^32 and (b) the
three-pass NTT (2^21 / 2^22 points) over random NTT-friendly primes, against the oracle.  Bit-exact.
Usage: python tools/fuzz_conv_ntt3.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _numtheory as nt
from oracle import gf_oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 777
rng = np.random.default_rng(seed)
t_end = time.time() + budget
n_conv = n_ntt = 0


def random_prime(lo, hi):
    while True:
        c = int(rng.integers(lo, hi)) | 1
        if nt.is_prime(c):
            return c


while time.time() < t_end:
    if rng.random() < 0.75:
        p = random_prime(3, 2**int(rng.integers(3, 33)))
        if p >= 2**32:
            continue
        GF = ga.GF(p)
        F = O.OracleField(p, 1, None, GF._primitive_element_int)
        na = int(rng.integers(64, 40000))
        nb = int(rng.integers(max(64, (1 << 22) // na + 1), max(65, (1 << 22) // na + 2, min(40000, 3 * 10**8 // na))))
        a = rng.integers(0, p, na, dtype=np.uint64)
        b = rng.integers(0, p, nb, dtype=np.uint64)
        if rng.random() < 0.3:
            a[: na // 2] = p - 1; b[: nb // 2] = p - 1
        dt = GF.dtypes[int(rng.integers(0, len(GF.dtypes)))]
        z = np.convolve(GF(a.astype(np.int64), dtype=dt), GF(b.astype(np.int64), dtype=dt))
        assert np.array_equal(z.numpy().astype(np.uint64), F.convolve(a, b)), ("convolve", p, na, nb, dt)
        n_conv += 1
    else:
        lg = int(rng.choice([21, 21, 22]))
        if rng.random() < 0.5:  # 32-bit: k * 2^lg + 1 < 2^32
            while True:
                p = int(rng.integers(1, 2**(32 - lg))) * 2**lg + 1
                if nt.is_prime(p):
                    break
        else:
            while True:
                p = int(rng.integers(2**20, 2**40)) * 2**lg + 1
                if p < 2**63 and nt.is_prime(p):
                    break
        GF = ga.GF(p)
        F = O.OracleField(p, 1, None, GF._primitive_element_int)
        n = 1 << lg
        x = rng.integers(0, p, n, dtype=np.uint64)
        X = np.fft.fft(GF(x.astype(np.int64)))
        assert np.array_equal(X.numpy().astype(np.uint64), F.ntt(x, omega=GF._root_of_unity_int(n))), ("ntt", p, lg)
        assert np.array_equal(np.fft.ifft(X).numpy().astype(np.uint64), x), ("intt", p, lg)
        n_ntt += 1
print(f"fuzz_conv_ntt3: {n_conv} CRT convolutions, {n_ntt} three-pass transforms, all identical to the oracle (seed {seed}, {budget:.0f} s)")
