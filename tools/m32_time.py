"""Timing of the signed-Montgomery NTT kernels (gfa_ntt_m32.hip) against the round-2 register kernels.

    python tools/m32_time.py            # 2^20 x 64 over GF(7340033) and a few other shapes, HIP events (gfa_time_ntt)
(The r03 tuning knobs are gone: their winners are hard-coded, the sweep is profiles/r03_m32_sweep.txt.)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
shapes = [(7340033, 20, 64), (7340033, 20, 256), (7340033, 16, 1024), (7340033, 10, 65536), (28311553, 20, 64), (65537, 12, 16384)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for p, logn, batch in shapes:
    P = ga.GF(p); N = 1 << logn
    x = torch.from_numpy(np.random.default_rng(3).integers(0, p, (batch, N), dtype=np.uint32).view(np.int32)).cuda()
    o = torch.empty_like(x)
    ms = ctypes.c_float()
    best = 1e9
    for _ in range(3):
        L.check(lib.gfa_time_ntt(P._handle, x.data_ptr(), o.data_ptr(), N, batch, P._root_of_unity_int(N), L.U32, st, 20, ctypes.byref(ms)))
        best = min(best, ms.value)
    frac = 8.0 * N * batch / (best * 1e-3) / 8e12
    print(f"p={p} 2^{logn} x {batch}: {best:.4f} ms   {frac:.3f} of the 8 B/point HBM roofline", flush=True)
