"""Tuning sweep of the signed-Montgomery NTT kernels (gfa_ntt_m32.hip) through gfa_debug_m32_tune:
  * three-pass form (2^21 .. 2^26 points): line lengths of the passes, non-temporal last pass on / off
  * two-pass 2^20 x 64: last pass in 1024-thread workgroups (32 lines per tile: whole 128-byte lines per transposed store)
Every configuration is checked against the default configuration's output before it is timed."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L

lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(3)


def timed(GF, x, o, n, batch, omega):
    ms = ctypes.c_float()
    L.check(lib.gfa_time_ntt(GF._handle, x.data_ptr(), o.data_ptr(), n, batch, omega, L.U32, st, 5, ctypes.byref(ms)))
    return ms.value


def run(GF, x, o, n, batch, omega):
    L.check(lib.gfa_ntt(GF._handle, x.data_ptr(), o.data_ptr(), n, batch, omega, 0, L.U32, st))
    torch.cuda.synchronize()


for p, lg in ((469762049, 26), (469762049, 24), (167772161, 22), (23068673, 21)):
    GF = ga.GF(p)
    n = 1 << lg
    x = torch.from_numpy(rng.integers(0, p, n, dtype=np.uint32).view(np.int32)).cuda()
    o = torch.empty_like(x)
    ref = torch.empty_like(x)
    omega = GF._root_of_unity_int(n)
    for k in range(4):
        lib.gfa_debug_m32_tune(k, -1 if k == 2 else 0)
    run(GF, x, ref, n, 1, omega)
    print(f"p={p} n=2^{lg}: default {timed(GF, x, o, n, 1, omega):.4f} ms  ({8.0 * n / 1e6 / timed(GF, x, o, n, 1, omega) / 8000:.4f} of the roofline)", flush=True)
    for l1 in range(5, 11):
        for l2 in range(5, 11):
            l3 = lg - l1 - l2
            if not 5 <= l3 <= 10 or (l1 + l2 + l3 > 24 and min(l1, l2, l3) < 7):
                continue
            for nt in (0, 1):
                lib.gfa_debug_m32_tune(0, l1); lib.gfa_debug_m32_tune(1, l2); lib.gfa_debug_m32_tune(2, nt)
                run(GF, x, o, n, 1, omega)
                ok = torch.equal(o, ref)
                ms = timed(GF, x, o, n, 1, omega)
                print(f"   split {l1:2d} {l2:2d} {l3:2d} nt={nt}: {ms:.4f} ms  {8.0 * n / 1e6 / ms / 8000:.4f} {'ok' if ok else 'MISMATCH'}", flush=True)
    del x, o, ref

for k in range(4):
    lib.gfa_debug_m32_tune(k, -1 if k == 2 else 0)
for p in (7340033, 469762049):
    GF = ga.GF(p)
    n, batch = 1 << 20, 64
    x = torch.from_numpy(rng.integers(0, p, (batch, n), dtype=np.uint32).view(np.int32)).cuda()
    o = torch.empty_like(x); ref = torch.empty_like(x)
    omega = GF._root_of_unity_int(n)
    run(GF, x, ref, n, batch, omega)
    for wide_wg in (0, 1):
        for nt in (-1, 0, 1):
            lib.gfa_debug_m32_tune(3, wide_wg); lib.gfa_debug_m32_tune(2, nt)
            run(GF, x, o, n, batch, omega)
            ok = torch.equal(o, ref)
            ms = timed(GF, x, o, n, batch, omega)
            print(f"p={p} 2^20 x 64: last pass {'1024' if wide_wg else ' 512'} threads, nt={nt:2d}: {ms:.4f} ms  {8.0 * n * batch / 1e6 / ms / 8000:.4f} {'ok' if ok else 'MISMATCH'}", flush=True)
    for k in range(4):
        lib.gfa_debug_m32_tune(k, -1 if k == 2 else 0)
