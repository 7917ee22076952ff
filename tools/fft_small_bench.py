"""The reference's own FFT benchmark sizes (benchmarks/test_fft.py:13-23): 256*K points, K = 1..9, over the first prime-power
field of order k*size + 1; wall-clock per np.fft.fft call through the Python front end, and 4096 transforms in one launch."""
import itertools, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd._ntt import fft_batched
for K in range(1, 10):
    size = 256 * K
    for order in itertools.count(size + 1, step=size):
        try:
            p, m = ga._numtheory.prime_power(order)
            break
        except Exception:
            continue
    GF = ga.GF(p, m)
    x = GF.Random(size, seed=K)
    np.fft.fft(x); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        y = np.fft.fft(x)
    torch.cuda.synchronize()
    single = (time.perf_counter() - t0) / 200 * 1e6
    xb = GF.Random((4096, size), seed=K)
    fft_batched(xb); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fft_batched(xb)
    torch.cuda.synchronize()
    batched = (time.perf_counter() - t0) / 10 * 1e6
    print(f"size {size:5d} over {GF.name:12s}: {single:7.1f} us per call; 4096 transforms in one launch: {batched:8.1f} us = {4096 / batched:.1f} M transforms/s")
