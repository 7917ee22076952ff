"""The README's example, executed (kept in sync by hand)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, galois_amd as galois
GF = galois.GF(2**8)
x, y = GF.Random(10**8, seed=1), GF.Random(10**8, seed=2)
z = x * y
rs = galois.ReedSolomon(255, 223)
c = rs.encode(GF.Random((1 << 17, 223)))
m, n_err = rs.decode(c, errors=True)
assert (n_err == 0).all()
X = np.fft.fft(galois.GF(7340033).Random(1 << 20))
Y = np.fft.fft(galois.GF(2**64 - 2**32 + 1).Random(1 << 26))
P = galois.GF(2**31 - 1)
c = np.convolve(P.Random(1 << 20), P.Random(1 << 20))
T = galois.GF(3**7); t = T.Random(10**7) / T.Random(10**7, low=1)
big = galois.GF(2**100); w = big(3) ** (2**99 + 12345)
from galois_amd._ntt import fft_batched
F = fft_batched(galois.GF(65537).Random((64, 1 << 16)))
bch = galois.BCH(1023, d=21)
assert (bch.n, bch.k, bch.t) == (1023, 923, 10)
A = galois.GF(251).Random((4096, 4096)); B = A @ np.linalg.inv(A)
E = galois.GF(7**7); q7 = E.Random(10**7) / E.Random(10**7, low=1)
assert bool(((q7 * 0) == E(0)).all())
c8 = np.convolve(galois.GF(2**8).Random(1 << 18), galois.GF(2**8).Random(1 << 18))
assert c8.shape == ((1 << 19) - 1,)
G = galois.GF(2**8).Random((2048, 2048)); H = G @ G
I8 = galois.GF(2**8).Identity(2048)
assert bool(((G @ I8) == G).all())
s8 = np.add.reduce(GF.Random(10**8, seed=3)); cs = np.add.accumulate(GF.Random(10**7, seed=4))
assert int(cs[-1]) == int(np.add.reduce(GF.Random(10**7, seed=4)))
print("readme example OK", z.shape, X.shape, Y.shape, c.shape, bool((B == galois.GF(251).Identity(4096)).all()))
