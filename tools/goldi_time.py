"""Goldilocks transforms: batched 2^20 x 16, single 2^24 / 2^26, and the C5 per-rank column / row passes (HIP events)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import galois_amd as ga
from galois_amd import _lib as L
lib = L.lib()
st = torch.cuda.current_stream().cuda_stream
p = 2**64 - 2**32 + 1
P = ga.GF(p)
ms = ctypes.c_float()
for logn, batch in ((10, 16384), (20, 16), (24, 1), (26, 1)):
    N = 1 << logn
    x = torch.empty((batch, N), dtype=torch.int64, device="cuda").random_(0, 2**62)
    o = torch.empty_like(x)
    L.check(lib.gfa_time_ntt(P._handle, x.data_ptr(), o.data_ptr(), N, batch, P._root_of_unity_int(N), L.U64, st, 10, ctypes.byref(ms)))
    pts = batch * N
    print(f"2^{logn} x {batch}: {ms.value:.4f} ms  {16 * pts / ms.value / 1e6:.0f} GB/s algorithmic  frac {16 * pts / ms.value / 1e6 / 8000:.3f}")
