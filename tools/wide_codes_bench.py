"""Throughput of the table-driven code kernels (gfa_rs_wide.hip): codes whose syndrome field has more than 256 elements."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga

rng = np.random.default_rng(5)


def timed(fn, reps=10, trials=3):
    """Seconds per call on the GPU's own clock: events on the current stream around `reps` back-to-back calls, best of `trials`
    (the first version timed the host wall clock over 3 calls and spread by 40 % run to run)."""
    fn(); torch.cuda.synchronize()
    best = float("inf")
    for _ in range(trials):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e-3)
    return best


for name, code, q in (("RS(1023,1003)/GF(2^10)", ga.ReedSolomon(1023, 1003, field=ga.GF(2**10)), 2**10),
                      ("BCH(1023,d=21)/GF(2)", ga.BCH(1023, d=21), 2),
                      ("RS(4095,4063)/GF(2^12)", ga.ReedSolomon(4095, 4063, field=ga.GF(2**12)), 2**12)):
    n, k, t = code.n, code.k, code.t
    N = 1 << 14
    M = code.field(rng.integers(0, q, (N, k)))
    C = code.encode(M)
    R = C.numpy().astype(np.int64)
    for i in range(N):
        ne = int(rng.integers(0, t + 1))
        pos = rng.choice(n, ne, replace=False)
        R[i, pos] = (R[i, pos] + rng.integers(1, q, ne)) % q
    Rg = code.field(R)
    sym = Rg._t.element_size()
    te = timed(lambda: code.encode(M))
    td = timed(lambda: code.decode(Rg, output="codeword"))
    tc = timed(lambda: code.decode(C, output="codeword"))
    ok = bool((code.decode(Rg, output="codeword") == C).all())
    print(f"{name}: k={k} t={t} {N} codewords  encode {te*1e3:.2f} ms ({N*n*sym/te/1e9:.2f} GB/s)  decode e~U{{0..t}} {td*1e3:.2f} ms "
          f"({N*n*sym/td/1e9:.2f} GB/s, {N/td/1e3:.0f} k codewords/s)  clean {tc*1e3:.2f} ms  ok={ok}", flush=True)
