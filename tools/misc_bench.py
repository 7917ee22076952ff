"""Survey of the array functions beside the element-wise ufuncs: reductions, scans, np.convolve, polynomial evaluation, np.log, np.sqrt --
wall time of the Python call (device-synchronised), element rate and fraction of 8 TB/s at the bytes the call has to move."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


for q, n in ((2**8, 10**8), (7340033, 5 * 10**7), (3**5, 10**8), (2**16, 5 * 10**7), (2**64 - 2**32 + 1, 2 * 10**7)):
    GF = ga.GF(q)
    x = GF.Random(n, seed=1)
    y = GF.Random(n, low=1, seed=2)
    es = x.numpy().itemsize if q < 2**63 else 8
    row = {"field": GF.name, "n": n}
    for name, fn, byt in (("add.reduce", lambda: np.add.reduce(x), es), ("multiply.reduce", lambda: np.multiply.reduce(y), es),
                          ("add.accumulate", lambda: np.add.accumulate(x), 2 * es), ("multiply.accumulate", lambda: np.multiply.accumulate(y), 2 * es)):
        t = timed(fn)
        row[name] = f"{n / t / 1e9:.0f} Gel/s ({byt * n / t / 8e12:.2f})"
    m = n // 10
    z = y[:m]
    t = timed(lambda: np.sqrt(z * z), 3)
    row["sqrt(z*z)"] = f"{m / t / 1e9:.1f} Gel/s"
    if q <= 2**32:
        t = timed(lambda: np.log(z), 3)
        row["log"] = f"{m / t / 1e9:.2f} Gel/s"
    p = ga.Poly(GF.Random(101, seed=3))
    t = timed(lambda: p(z), 3)
    row["poly deg 100"] = f"{m / t / 1e9:.2f} Gpt/s = {100 * m / t / 1e12:.2f} TMAC/s"
    a, b = GF.Random(4096, seed=4), GF.Random(4096, seed=5)
    t = timed(lambda: np.convolve(a, b), 5)
    row["convolve 4096x4096"] = f"{t * 1e6:.0f} us"
    a, b = GF.Random(1 << 18, seed=4), GF.Random(1 << 18, seed=5)
    t = timed(lambda: np.convolve(a, b), 3)
    row["convolve 2^18 x 2^18"] = f"{t * 1e3:.2f} ms"
    print(row, flush=True)
