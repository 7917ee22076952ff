"""np.convolve over GF(2^31 - 1) (no power-of-two roots: direct kernel or three NTT primes + CRT inside gfa_convolve).
(The route is chosen by size inside gfa_convolve: CRT from 2^22 multiply-adds; the crossover was measured with knobs that r04 removed.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga

GF = ga.GF(int(os.environ.get("CONV_P", 2**31 - 1)))
rng = np.random.default_rng(3)
sizes = [int(s) for s in sys.argv[1:]] or [256, 1024, 4096, 16384, 65536]
for n in sizes:
    for nb in sorted({n, max(64, n // 16)}):
        a = GF(rng.integers(0, GF.order, n, dtype=np.int64)); b = GF(rng.integers(0, GF.order, nb, dtype=np.int64))
        np.convolve(a, b); torch.cuda.synchronize()
        reps = 5 if n * nb < 2**32 else 1
        t = time.perf_counter()
        for _ in range(reps):
            c = np.convolve(a, b)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        print(f"{n:>8} x {nb:>8}: {dt * 1e3:9.3f} ms  ({n * nb / dt / 1e9:8.1f} G mul-add/s equivalent)", flush=True)
